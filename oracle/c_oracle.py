"""
ctypes wrapper around oracle/_build/liboracle*.so (oracle/loopnest.c).
TEST INFRASTRUCTURE ONLY -- see oracle/np_oracle.py.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_lib = None


def build(force: bool = False) -> None:
    """gcc -O3 -fopenmp oracle/loopnest.c -> oracle/_build/liboracle_generic.so (portable)."""
    if force:
        subprocess.run(["make", "-C", str(_DIR), "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(_DIR), "all"], check=True, capture_output=True)


def build_native() -> Path | None:
    """-march=native build for THIS machine, in its temp dir (never travels)."""
    import os
    import shutil
    import tempfile

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        return None
    d = Path(tempfile.gettempdir()) / f"feinsum_oracle_{os.getuid()}"
    d.mkdir(exist_ok=True)
    out = d / "liboracle_native.so"
    res = subprocess.run([cc, "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=c11",
                          "-o", str(out), str(_DIR / "loopnest.c")], capture_output=True)
    return out if res.returncode == 0 else None


def _typed(lib):
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    for fam in ("grad3d", "div3d"):
        for kind in ("trivial", "hoisted"):
            fn = getattr(lib, f"oracle_{fam}_{kind}")
            fn.restype = None
            fn.argtypes = [dp, dp, dp, dp, C.c_int64, C.c_int]
    for kind in ("trivial", "hoisted"):
        fn = getattr(lib, f"oracle_facemass_{kind}")
        fn.restype = None
        fn.argtypes = [dp, dp, dp, dp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_set_num_threads.restype = None
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    return lib


def load(native: bool = False):
    """The portable build (default) or, for timing, the machine-local native one."""
    global _lib
    if native:
        p = build_native()
        if p is not None:
            return _typed(C.CDLL(str(p)))
    if _lib is None:
        p = _DIR / "_build" / "liboracle_generic.so"
        if not p.exists():
            build()
        _lib = _typed(C.CDLL(str(p)))
    return _lib


def grad3d(J, D, u, kind="hoisted"):
    E, Np = u.shape
    out = np.empty((3, E, Np))
    getattr(load(), f"oracle_grad3d_{kind}")(J, D, u, out, E, Np)
    return out


def div3d(J, D, u, kind="hoisted"):
    _, E, Np = u.shape
    out = np.empty((E, Np))
    getattr(load(), f"oracle_div3d_{kind}")(J, D, u, out, E, Np)
    return out


def facemass(J, R, v, kind="hoisted", jfe=False, rifj=False):
    nf, E, Nfp = v.shape
    Np = R.shape[0] if rifj else R.shape[1]
    out = np.empty((E, Np))
    getattr(load(), f"oracle_facemass_{kind}")(J, R, v, out, E, Np, nf, Nfp, int(jfe), int(rifj))
    return out


def num_threads() -> int:
    return int(load().oracle_num_threads())
