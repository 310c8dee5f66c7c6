"""ctypes binding of ``csrc/libddmp_hip.so``.

The prototypes are parsed from ``include/ddmp_hip.h`` (the C ABI is the contract; this file adds
nothing to it).  There is NO CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the HIP path raises.

``build()`` compiles the library in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("DDMP_LIB") or os.path.join(CSRC, "libddmp_hip.so")   # DDMP_LIB: diagnostic builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ddmp_hip.h")

_SCALARS = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "float": ctypes.c_float,
    "double": ctypes.c_double, "size_t": ctypes.c_size_t, "ddmp_stream": ctypes.c_void_p,
}
_RET = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}


class DdmpError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype_key, [(ctype, argname), ...])} for every ``ddmp_*`` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const\s+char\s*\*)\s+(ddmp_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret = re.sub(r"\s+", " ", m.group(1)).replace(" *", "*")
        name, args = m.group(2), m.group(3).strip()
        sig = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    sig.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    toks = [t for t in a.split(" ") if t != "const"]
                    sig.append((_SCALARS[toks[0]], toks[-1]))
        protos[name] = (ret, sig)
    return protos


def build(verbose: bool = False) -> str:
    """``make -C csrc`` (hipcc --offload-arch=gfx950).  Returns the library path."""
    r = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise DdmpError("building libddmp_hip.so failed (hipcc for gfx950):\n" + r.stdout[-4000:])
    return LIB_PATH


_lib = None
_protos = None


def lib():
    """Load (once) and return the ctypes library with argtypes/restypes set from the header."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DdmpError(
            "HIP extension not built: %s is missing.  Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "or `make -C dual-dmp_amd/csrc`.  There is no CPU fallback for the product path." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64: load it FIRST so that libddmp_hip.so binds to the same HIP runtime (two
    # runtimes in one process do not share devices or streams: hipMalloc then fails with "no ROCm-capable device")
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DdmpError("cannot load %s: %s" % (LIB_PATH, e))
    _protos = parse_header()
    for name, (ret, sig) in _protos.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise DdmpError("libddmp_hip.so does not export %s (declared in include/ddmp_hip.h)" % name)
        fn.restype = _RET[ret]
        fn.argtypes = [t for t, _ in sig]
    if handle.ddmp_abi_version() != 3:
        raise DdmpError("libddmp_hip.so ABI version mismatch")
    abl = handle.ddmp_build_ablation_flags()
    if abl and not os.environ.get("DDMP_LIB"):
        raise DdmpError("%s is a timing-only ablation build (flags %d: parts of its kernels are compiled out, results are WRONG); "
                        "rebuild with `make -C dual-dmp_amd/csrc clean all`, or name a diagnostic build explicitly through DDMP_LIB"
                        % (LIB_PATH, abl))
    _lib = handle
    return _lib


def status_string(st: int) -> str:
    s = lib().ddmp_status_string(int(st))
    return s.decode() if s else "?"


def check(st: int, what: str = ""):
    if st != 0:
        raise DdmpError("%s failed: status %d (%s)" % (what or "ddmp call", st, status_string(st)))
