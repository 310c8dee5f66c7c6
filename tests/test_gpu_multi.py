"""One process per GPU over RCCL against the single-device trainer ("a 2-GPU torchrun nccl parity test ...
covering both interleave settings").  Uses min(2, visible GPUs) ranks: on a one-GPU box this is the RCCL path at world
size 1 (communicator, all_to_all_single with empty splits, all-reduces, ghost exchanges); on a multi-GPU box the halos,
ghost rows and partial sums really travel."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind,losses,interleave,native,streams,extra", [
    ("ico4", "sharded", "0", "0", "1", {}), ("grid", "sharded", "1", "0", "1", {}), ("ico4", "replicated", "1", "0", "1", {}),
    # the library's own RCCL communicators (the default backend): PosNet on a second stream with its own communicator, and
    # the single-stream form
    ("grid", "sharded", "0", "1", "1", {}), ("ico4", "replicated", "0", "1", "1", {}), ("grid", "sharded", "0", "1", "0", {}),
    # round 4: at world size 1 every exchange still goes THROUGH RCCL (DDMP_COMM_LOOPBACK=1: a self send/recv inside the group
    # and a one-rank all-reduce per exchange), eagerly and with the whole partitioned iteration captured into one hipGraph
    # (DDMP_DIST_GRAPH=1: eager, capture, replay), on one stream and on two
    ("grid", "sharded", "0", "1", "0", {"DDMP_COMM_LOOPBACK": "1"}),
    ("grid", "sharded", "0", "1", "1", {"DDMP_COMM_LOOPBACK": "1"}),
    ("grid", "sharded", "0", "1", "0", {"DDMP_COMM_LOOPBACK": "1", "DDMP_DIST_GRAPH": "1"}),
    ("ico4", "sharded", "0", "1", "1", {"DDMP_COMM_LOOPBACK": "1", "DDMP_DIST_GRAPH": "1"}),
    # round 6: one rank WITHOUT the loopback issues no RCCL call: there the capture forks PosNet's stream (RCCL operations on a
    # forked stream end the capture in a SIGSEGV on ROCm 7.2, with one communicator as with two: profiles/r06_dist_overhead.txt)
    ("grid", "sharded", "0", "1", "1", {"DDMP_DIST_GRAPH": "1"}),
    # round 6: the communicator's EXCHANGE STREAM (every RCCL call on a stream of its own behind an event of the kernels' stream;
    # what the interior / boundary overlap rides on) with every call really going through RCCL
    ("grid", "sharded", "0", "1", "0", {"DDMP_COMM_LOOPBACK": "1", "DDMP_DIST_SPLIT": "1"}),
    ("ico4", "sharded", "0", "1", "1", {"DDMP_COMM_LOOPBACK": "1", "DDMP_DIST_SPLIT": "1"})])
def test_rccl_ranks_match_single_device(kind, losses, interleave, native, streams, extra):
    n = min(2, torch.cuda.device_count())                   # counting devices does not initialise the GPU in this process
    assert n >= 1
    if extra and n > 1:
        pytest.skip("the loopback switch is for one-rank communicators")
    env = dict(os.environ, DDMP_DIST_INTERLEAVE=interleave, DDMP_DIST_NATIVE=native, DDMP_DIST_STREAMS=streams,
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    port = 29600 + (os.getpid() + hash((kind, losses, interleave, native, streams, tuple(sorted(extra))))) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "nccl_worker.py"), kind, losses, "4" if extra.get("DDMP_DIST_GRAPH") else "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300 if extra else 600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "PARITY" in r.stdout, tail
    if extra.get("DDMP_DIST_GRAPH"):
        two = streams == "1" and "DDMP_COMM_LOOPBACK" not in extra
        assert "captured=1" in r.stdout and ("streams=%d" % (2 if two else 1)) in r.stdout, tail
