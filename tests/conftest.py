import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (small meshes) is run by PyTorch-CPU: on a many-core GPU host the all-cores default is tens of
    # times slower than a handful of threads (measured: 40x on a 256-core box; one suite run took 523 s instead of 84 s)
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 8))
    except Exception:  # pragma: no cover
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_oracle():
    """The oracle is test infrastructure: imported by path, never by the product package."""
    import importlib.util
    p = os.path.join(ROOT, "oracle", "ddmp_oracle.py")
    spec = importlib.util.spec_from_file_location("ddmp_oracle", p)
    mod = sys.modules.get("ddmp_oracle")
    if mod is None:
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ddmp_oracle"] = mod
        spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


def pytest_collection_finish(session):
    """The two 144k-face teacher-forced comparisons (tests/test_gpu_path.py) need 50-100 s of CPU oracle each while the GPU idles:
    when they are selected, their oracle halves start now in two worker processes and run beside the rest of the suite."""
    if any("teacher_forced" in it.nodeid and "_at_144k" in it.nodeid for it in session.items) \
            and os.environ.get("DDMP_TEST_ORACLE_WORKERS", "1") != "0" and (os.cpu_count() or 1) >= 24:
        import oracle_jobs
        oracle_jobs.start_big_jobs()


def pytest_sessionfinish(session, exitstatus):
    mod = sys.modules.get("oracle_jobs")
    if mod is not None:
        mod.shutdown()
