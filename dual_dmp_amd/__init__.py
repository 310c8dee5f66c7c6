"""Import shim: the product package lives in the directory ``dual-dmp_amd/``
(a name Python cannot import directly).  ``import dual_dmp_amd`` resolves here,
re-points ``__path__`` at that directory and executes its ``__init__``; every
submodule (``dual_dmp_amd.mesh`` ...) is then loaded from ``dual-dmp_amd/``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dual-dmp_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
