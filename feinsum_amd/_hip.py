"""
ctypes binding of ``libfeinsum_hip.so`` (C ABI: ``include/feinsum_hip.h``).

This is the only place where the Python host code touches native code.  It
replaces the reference's PyOpenCL enqueue inside the loopy executor
(reference: ``src/feinsum/measure.py:163-165,243-251,267``).  There is no CPU
fallback: if the library is missing or fails to load, every entry point raises
:class:`~feinsum_amd.diagnostics.HipLibraryError`.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

from feinsum_amd.diagnostics import HipLibraryError, InvalidParameterError

FE_OK, FE_EINVAL, FE_EUNSUPPORTED, FE_EHIP = 0, -1, -2, -3
VARIANT_AUTO, VARIANT_GENERIC, VARIANT_MFMA, VARIANT_TILED, VARIANT_MFMA_SPLIT = 0, 1, 2, 3, 4
# "mfma_split": div of tetrahedra p = 1..4 only -- the MFMA kernel walking both halves of the element range at once
# (two write windows for its one output: include/feinsum_hip.h, FE_VARIANT_MFMA_SPLIT)
VARIANTS = {"auto": VARIANT_AUTO, "generic": VARIANT_GENERIC, "mfma": VARIANT_MFMA, "tiled": VARIANT_TILED,
            "mfma_split": VARIANT_MFMA_SPLIT}

#: every symbol declared in include/feinsum_hip.h (checked by the CPU test-suite)
EXPORTED_SYMBOLS = (
    "fe_version", "fe_last_error", "fe_device_count", "fe_device_info",
    "fe_grad3d_f64", "fe_div3d_f64", "fe_grad3d_f64_ex", "fe_div3d_f64_ex", "fe_divcomp3d_f64",
    "fe_grad3d_batched_f64", "fe_div3d_batched_f64", "fe_gradplanes3d_f64", "fe_matapply_f64",
    "fe_grad_f64", "fe_div_f64",
    "fe_graddiv3d_f64", "fe_waveop3d_f64",
    "fe_facemass_f64",
    "fe_flops_per_element", "fe_time_launches", "fe_einsum_generic", "fe_kernel_resources",
    "fe_prepare_operator", "fe_grad3d_prepared_f64", "fe_div3d_prepared_f64", "fe_facemass_prepared_f64",
    "fe_graddiv3d_prepared_f64", "fe_waveop3d_prepared_f64", "fe_divcomp_f64", "fe_release_prepared",
    "fe_split_alloc", "fe_split_free", "fe_split_info", "fe_split_stats", "fe_split_reserve", "fe_split_trim", "fe_launch_f32", "fe_set_tail_rounds", "fe_set_tail_min_rounds",
    "fe_set_cu_limit", "fe_set_phase_priority_p5", "fe_set_div_interleave", "fe_set_div_quarter_tail", "fe_set_grad_quarter_tail", "fe_set_grad_staggered_start", "fe_last_launch_info", "fe_stream_retired", "fe_capture_id", "fe_graph_retired", "fe_tail_stats", "fe_tail_check", "fe_tail_plant", "fe_set_temporal_loads_mib", "fe_set_write_through_mib",
)
FAMILY_F32 = 0x100    # FE_FAMILY_F32

_c_double_p = C.c_void_p   # device pointers travel as plain integers


class ArgPack(C.Structure):
    """``struct fe_argpack`` of include/feinsum_hip.h."""

    _fields_ = [
        ("J", C.c_void_p), ("D", C.c_void_p), ("u", C.c_void_p), ("v_div", C.c_void_p),
        ("out", C.c_void_p), ("out2", C.c_void_p),
        ("v", C.POINTER(C.c_void_p)), ("outs", C.POINTER(C.c_void_p)),
        ("E", C.c_int64),
        ("Np", C.c_int32), ("nf", C.c_int32), ("Nfp", C.c_int32), ("b", C.c_int32),
        ("layout_flags", C.c_int32), ("variant", C.c_int32),
        ("j3", C.POINTER(C.c_void_p)),
        ("ndim", C.c_int32),
        ("prepared", C.c_void_p),
    ]


FE_MAX_EINSUM_OPERANDS = 8
FE_MAX_EINSUM_INDICES = 8


class EinsumDesc(C.Structure):
    """``struct fe_einsum_desc`` of include/feinsum_hip.h."""

    _fields_ = [
        ("n_operands", C.c_int32), ("n_out", C.c_int32), ("n_sum", C.c_int32), ("dtype", C.c_int32),
        ("out_extent", C.c_int64 * FE_MAX_EINSUM_INDICES),
        ("sum_extent", C.c_int64 * FE_MAX_EINSUM_INDICES),
        ("op_out_stride", (C.c_int64 * FE_MAX_EINSUM_INDICES) * FE_MAX_EINSUM_OPERANDS),
        ("op_sum_stride", (C.c_int64 * FE_MAX_EINSUM_INDICES) * FE_MAX_EINSUM_OPERANDS),
    ]


def library_path() -> Path:
    """``$FEINSUM_HIP_LIB`` if set, else the in-tree ``feinsum_amd/libfeinsum_hip.so``."""
    env = os.environ.get("FEINSUM_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / "libfeinsum_hip.so"


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Load (once) and type the shared library; never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise HipLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950) or point FEINSUM_HIP_LIB at it")
    # torch's wheel carries its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7) and asks for it by FILE name:
    # loaded after this library -- which resolves libamdhip64.so.7 to /opt/rocm/lib -- it becomes a SECOND runtime in the process,
    # and launches here then fail with "no ROCm-capable device is detected" while torch owns the device.  Loaded first, its soname
    # satisfies this library too: one runtime.  (A process without torch -- the C ABI's other callers -- has only one anyway.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:
        raise HipLibraryError(f"cannot load {path}: {exc}") from exc

    lib.fe_version.restype = C.c_int
    lib.fe_version.argtypes = []
    lib.fe_last_error.restype = C.c_char_p
    lib.fe_last_error.argtypes = []
    lib.fe_device_count.restype = C.c_int
    lib.fe_device_count.argtypes = []
    lib.fe_device_info.restype = C.c_int
    lib.fe_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for name in ("fe_grad3d_f64", "fe_div3d_f64"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    for name in ("fe_grad3d_f64_ex", "fe_div3d_f64_ex", "fe_divcomp3d_f64"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    for name in ("fe_grad3d_batched_f64", "fe_div3d_batched_f64"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                       C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    for name in ("fe_grad_f64", "fe_div_f64"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                       C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_matapply_f64.restype = C.c_int
    lib.fe_matapply_f64.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_gradplanes3d_f64.restype = C.c_int
    lib.fe_gradplanes3d_f64.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_graddiv3d_f64.restype = C.c_int
    lib.fe_graddiv3d_f64.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_waveop3d_f64.restype = C.c_int
    lib.fe_waveop3d_f64.argtypes = [C.c_void_p] * 8 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                       C.c_int64] + [C.c_int32] * 6 + [C.c_void_p]
    lib.fe_facemass_f64.restype = C.c_int
    lib.fe_facemass_f64.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_divcomp_f64.restype = C.c_int
    lib.fe_divcomp_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_release_prepared.restype = C.c_int
    lib.fe_release_prepared.argtypes = [C.c_void_p]
    lib.fe_split_alloc.restype = C.c_int
    lib.fe_split_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int32]
    lib.fe_split_free.restype = C.c_int
    lib.fe_split_free.argtypes = [C.c_void_p]
    lib.fe_split_info.restype = C.c_int
    lib.fe_split_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.fe_split_stats.restype = C.c_int
    lib.fe_split_stats.argtypes = [C.c_char_p, C.c_size_t]
    lib.fe_split_reserve.restype = C.c_int
    lib.fe_split_reserve.argtypes = [C.c_size_t]
    lib.fe_split_trim.restype = C.c_int
    lib.fe_split_trim.argtypes = []
    lib.fe_set_tail_rounds.restype = C.c_int
    lib.fe_set_tail_rounds.argtypes = [C.c_int32]
    lib.fe_set_tail_min_rounds.restype = C.c_int
    lib.fe_set_tail_min_rounds.argtypes = [C.c_int32]
    lib.fe_set_temporal_loads_mib.restype = C.c_int
    lib.fe_set_temporal_loads_mib.argtypes = [C.c_int32]
    lib.fe_set_write_through_mib.restype = C.c_int
    lib.fe_set_write_through_mib.argtypes = [C.c_int32]
    lib.fe_set_cu_limit.restype = C.c_int
    lib.fe_set_cu_limit.argtypes = [C.c_int32]
    lib.fe_last_launch_info.restype = C.c_int
    lib.fe_last_launch_info.argtypes = [C.POINTER(C.c_int64), C.c_int32]
    lib.fe_set_phase_priority_p5.restype = C.c_int
    lib.fe_set_phase_priority_p5.argtypes = [C.c_int32]
    lib.fe_set_div_quarter_tail.restype = C.c_int
    lib.fe_set_div_quarter_tail.argtypes = [C.c_int32]
    lib.fe_set_grad_quarter_tail.restype = C.c_int
    lib.fe_set_grad_quarter_tail.argtypes = [C.c_int32]
    lib.fe_set_grad_staggered_start.restype = C.c_int
    lib.fe_set_grad_staggered_start.argtypes = [C.c_int32]
    lib.fe_set_div_interleave.restype = C.c_int64
    lib.fe_set_div_interleave.argtypes = [C.c_int64]
    lib.fe_stream_retired.restype = C.c_int
    lib.fe_stream_retired.argtypes = [C.c_void_p]
    lib.fe_tail_plant.restype = C.c_int
    lib.fe_tail_plant.argtypes = [C.c_void_p, C.c_uint32]
    lib.fe_capture_id.restype = C.c_int
    lib.fe_capture_id.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.fe_graph_retired.restype = C.c_int
    lib.fe_graph_retired.argtypes = [C.c_uint64]
    lib.fe_tail_stats.restype = C.c_int
    lib.fe_tail_stats.argtypes = [C.POINTER(C.c_int64), C.c_int32]
    lib.fe_tail_check.restype = C.c_int
    lib.fe_tail_check.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]
    lib.fe_launch_f32.restype = C.c_int
    lib.fe_launch_f32.argtypes = [C.c_int32, C.POINTER(ArgPack), C.c_void_p]
    lib.fe_prepare_operator.restype = C.c_int
    lib.fe_prepare_operator.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p]
    for name in ("fe_grad3d_prepared_f64", "fe_div3d_prepared_f64"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                       C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_facemass_prepared_f64.restype = C.c_int
    lib.fe_facemass_prepared_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_graddiv3d_prepared_f64.restype = C.c_int
    lib.fe_graddiv3d_prepared_f64.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    lib.fe_waveop3d_prepared_f64.restype = C.c_int
    lib.fe_waveop3d_prepared_f64.argtypes = ([C.c_void_p] * 10 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int64]
                                             + [C.c_int32] * 6 + [C.c_void_p])
    lib.fe_kernel_resources.restype = C.c_int
    lib.fe_kernel_resources.argtypes = [C.c_char_p, C.c_size_t]
    lib.fe_flops_per_element.restype = C.c_int64
    lib.fe_flops_per_element.argtypes = [C.c_int32] * 5
    lib.fe_time_launches.restype = C.c_int
    lib.fe_time_launches.argtypes = [C.c_int32, C.POINTER(ArgPack), C.c_int32, C.c_void_p,
                                     C.POINTER(C.c_float)]
    lib.fe_einsum_generic.restype = C.c_int
    lib.fe_einsum_generic.argtypes = [C.POINTER(EinsumDesc), C.POINTER(C.c_void_p), C.c_void_p,
                                      C.c_void_p]
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map a C return code to the Python exception the reference's callers expect."""
    if rc == FE_OK:
        return
    msg = (load_library().fe_last_error() or b"").decode("utf-8", "replace")
    if rc == FE_EINVAL:
        raise InvalidParameterError(msg)
    if rc == FE_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise HipLibraryError(f"HIP error ({rc}): {msg}")


def variant_code(variant) -> int:
    if variant is None:
        return VARIANT_AUTO
    if isinstance(variant, str):
        try:
            return VARIANTS[variant]
        except KeyError:
            raise InvalidParameterError(f"unknown kernel variant '{variant}'") from None
    return int(variant)


def device_info(dev: int = 0):
    """(name, peak fp64 GFLOP/s, peak GB/s) of HIP device *dev*."""
    lib = load_library()
    name = C.create_string_buffer(256)
    pf, pb = C.c_double(), C.c_double()
    check(lib.fe_device_info(dev, name, 256, C.byref(pf), C.byref(pb)))
    return name.value.decode(), pf.value, pb.value


def _ptr_array(ptrs: Sequence[int]):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr


OP_TRANSPOSED = 1   # FE_OP_TRANSPOSED: operator stored [3][Np(j)][Np(i)]
OP_J_ES = 2         # FE_OP_J_ES: div component, J stored [E][3]


def grad3d(J: int, D: int, u: int, out: int, E: int, Np: int, variant=None, stream: int = 0,
           op_flags: int = 0) -> None:
    check(load_library().fe_grad3d_f64_ex(J, D, u, out, E, Np, op_flags, variant_code(variant), stream))


def div3d(J: int, D: int, u: int, out: int, E: int, Np: int, variant=None, stream: int = 0,
          op_flags: int = 0) -> None:
    check(load_library().fe_div3d_f64_ex(J, D, u, out, E, Np, op_flags, variant_code(variant), stream))


def divcomp3d(J: int, D: int, u: int, out: int, E: int, Np: int, variant=None, stream: int = 0,
              op_flags: int = 0) -> None:
    check(load_library().fe_divcomp3d_f64(J, D, u, out, E, Np, op_flags, variant_code(variant), stream))


def grad3d_batched(J: int, D: int, u: Sequence[int], out: Sequence[int], E: int, Np: int,
                   variant=None, stream: int = 0, op_flags: int = 0) -> None:
    """``len(u)`` fields through one grad launch, sharing J and D."""
    if len(u) != len(out):
        raise InvalidParameterError("grad: need as many outputs as fields")
    check(load_library().fe_grad3d_batched_f64(J, D, _ptr_array(u), _ptr_array(out), E, Np, len(u),
                                               op_flags, variant_code(variant), stream))


def div3d_batched(J: int, D: int, u: Sequence[int], out: Sequence[int], E: int, Np: int,
                  variant=None, stream: int = 0, op_flags: int = 0) -> None:
    """``len(u)`` fields through one div launch, sharing J and D."""
    if len(u) != len(out):
        raise InvalidParameterError("div: need as many outputs as fields")
    check(load_library().fe_div3d_batched_f64(J, D, _ptr_array(u), _ptr_array(out), E, Np, len(u),
                                              op_flags, variant_code(variant), stream))


def gradplanes3d(J3: Sequence[int], D: int, u: Sequence[int], out: Sequence[Optional[int]], E: int,
                 Np: int, variant=None, stream: int = 0, op_flags: int = 0) -> None:
    """Rows of a batched 're,rij,ej->ei' sharing u and D: ``out[3k + x]`` from ``J3[x]`` and ``u[k]``
    (``None`` = plane not wanted)."""
    if len(J3) != 3 or len(out) != 3 * len(u):
        raise InvalidParameterError("grad planes: need 3 J arrays and 3 output slots per field")
    check(load_library().fe_gradplanes3d_f64(_ptr_array(J3), D, _ptr_array(u), _ptr_array(out), E, Np,
                                             len(u), op_flags, variant_code(variant), stream))


def matapply(J: Optional[int], D: int, u: Sequence[int], out: Sequence[int], E: int, Np: int,
             variant=None, stream: int = 0, op_flags: int = 0) -> None:
    """``out_k[e,i] = J[e] sum_j D[i,j] u_k[e,j]`` for ``len(u)`` fields (``J = None``: no factor)."""
    if len(u) != len(out):
        raise InvalidParameterError("matapply: need as many outputs as fields")
    check(load_library().fe_matapply_f64(J, D, _ptr_array(u), _ptr_array(out), E, Np, len(u), op_flags,
                                         variant_code(variant), stream))


def graddiv3d(J: int, D: int, u_grad: int, v_div: int, grad_out: int, div_out: int, E: int,
              Np: int, variant=None, stream: int = 0) -> None:
    check(load_library().fe_graddiv3d_f64(J, D, u_grad, v_div, grad_out, div_out, E, Np,
                                          variant_code(variant), stream))


def waveop3d(J: int, D: int, u_grad: int, grad_out: int, v_div: int, div_out: int, Jface: int, R: int,
             f: Sequence[int], lift: Sequence[int], E: int, Np: int, nf: int, Nfp: int,
             fm_layout_flags: int = 0, variant=None, stream: int = 0) -> None:
    """div(v), grad(u) and the lift of ``len(f)`` face fields in one persistent launch."""
    if len(f) != len(lift):
        raise InvalidParameterError("waveop: need as many lift outputs as face fields")
    check(load_library().fe_waveop3d_f64(J, D, u_grad, grad_out, v_div, div_out, Jface, R,
                                         _ptr_array(f), _ptr_array(lift), E, Np, nf, Nfp, len(f),
                                         fm_layout_flags, variant_code(variant), stream))


def facemass(J: int, R: int, v: Sequence[int], out: Sequence[int], E: int, Np: int, nf: int,
             Nfp: int, layout_flags: int = 0, variant=None, stream: int = 0) -> None:
    if len(v) != len(out):
        raise InvalidParameterError("face-mass: need as many outputs as fields")
    check(load_library().fe_facemass_f64(J, R, _ptr_array(v), _ptr_array(out), E, Np, nf, Nfp,
                                         len(v), layout_flags, variant_code(variant), stream))


PREPARED_OPERATOR_BYTES = 96 * 1024   # FE_PREPARED_OPERATOR_BYTES


def prepare_operator(family: int, op: int, Np: int, nf: int, Nfp: int, flags: int, prepared: int,
                     stream: int = 0) -> None:
    """Write operator *op* in the kernels' fragment layout into the device buffer *prepared*
    (PREPARED_OPERATOR_BYTES bytes).  NotImplementedError: this shape has no prepared form."""
    check(load_library().fe_prepare_operator(family, op, Np, nf, Nfp, flags, prepared, stream))


def release_prepared(prepared: int) -> None:
    """Drop the library's record of a prepared-operator buffer (before the buffer is freed)."""
    check(load_library().fe_release_prepared(prepared))


def split_alloc(nbytes: int) -> int:
    """Device pointer of a new array of the split allocator on the CURRENT device (fe_split_alloc)."""
    ptr = C.c_void_p()
    check(load_library().fe_split_alloc(C.byref(ptr), nbytes, 0))
    return int(ptr.value)


def split_free(ptr: int) -> None:
    check(load_library().fe_split_free(ptr))


def _json_call(fn, *args) -> dict:
    import json

    buf = C.create_string_buffer(1 << 14)
    n = fn(*args, buf, len(buf))
    if n < 0:
        check(n)
    return json.loads(buf.value.decode())


def split_info(ptr: int) -> dict:
    """What the split allocator did for one array: bytes, mapped bytes, class of every piece, milliseconds."""
    return _json_call(load_library().fe_split_info, ptr)


def split_stats() -> dict:
    """The current device's pool of the split allocator."""
    return _json_call(load_library().fe_split_stats)


def split_reserve(nbytes: int) -> None:
    check(load_library().fe_split_reserve(C.c_size_t(int(nbytes))))


def split_trim() -> None:
    check(load_library().fe_split_trim())


def kernel_resources() -> str:
    """Registers / LDS / resident blocks per CU of the kernels configured so far in this process."""
    buf = C.create_string_buffer(1 << 16)
    n = load_library().fe_kernel_resources(buf, len(buf))
    if n < 0:
        check(n)
    return buf.value.decode()


def flops_per_element(family: int, Np: int, nf: int = 0, Nfp: int = 0, b: int = 1) -> int:
    return int(load_library().fe_flops_per_element(family, Np, nf, Nfp, b))


def time_launches(family: int, pack: ArgPack, n_launches: int, stream: int = 0) -> float:
    """Milliseconds for *n_launches* back-to-back launches (HIP events on *stream*)."""
    ms = C.c_float()
    check(load_library().fe_time_launches(family, C.byref(pack), n_launches, stream, C.byref(ms)))
    return float(ms.value)


def einsum_generic(desc: EinsumDesc, operands: Sequence[int], out: int, stream: int = 0) -> None:
    check(load_library().fe_einsum_generic(C.byref(desc), _ptr_array(operands), out, stream))


def set_tail_rounds(rounds: int) -> int:
    """Number of dynamic rounds at the end of the persistent walks (fe_set_tail_rounds; negative: static walk); returns the
    previous value.  A tuning knob -- results do not depend on it."""
    return int(load_library().fe_set_tail_rounds(int(rounds)))


def set_tail_min_rounds(rounds: int) -> int:
    """Launches of fewer than *rounds* full rounds walk statically (fe_set_tail_min_rounds; default 4); returns the previous
    setting.  A tuning knob -- results do not depend on it."""
    return int(load_library().fe_set_tail_min_rounds(int(rounds)))


def set_cu_limit(cus: int) -> int:
    """Size the persistent grids as if the device had *cus* compute units (fe_set_cu_limit; 0 = the device's own count);
    returns the previous limit.  Results do not depend on it."""
    return int(load_library().fe_set_cu_limit(int(cus)))


def stream_retired(stream: int) -> bool:
    """Tell the library that *stream* was destroyed, so that its ticket-counter group can serve another stream."""
    rc = int(load_library().fe_stream_retired(stream))
    if rc < 0:
        check(rc)
    return rc == 1


def tail_check(repair: bool = False) -> dict:
    """Wait for the device and verify that every ticket counter is zero (fe_tail_check): ``dirty_words`` must be 0."""
    dirty, groups, streams, captured = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    check(load_library().fe_tail_check(int(bool(repair)), C.byref(dirty), C.byref(groups), C.byref(streams),
                                       C.byref(captured)))
    return {"dirty_words": int(dirty.value), "groups": int(groups.value), "streams": int(streams.value),
            "captured": int(captured.value)}


_TAIL_STATS = ("groups", "streams", "captured", "spare", "exhausted", "static_fallbacks", "grow_failures", "verified_after_error",
               "repaired_after_error", "live_captures", "max_groups", "hip_errors")


def tail_stats() -> dict:
    """Counters of the ticket-counter pool of the current device (fe_tail_stats): ``exhausted`` / ``static_fallbacks`` say how
    many launches that wanted tickets walked statically instead."""
    buf = (C.c_int64 * len(_TAIL_STATS))()
    n = int(load_library().fe_tail_stats(buf, len(_TAIL_STATS)))
    if n < 0:
        check(n)
    return {k: int(buf[i]) for i, k in enumerate(_TAIL_STATS[:n])}


def capture_id(stream: int) -> int:
    """Id of the stream capture in progress on *stream* (fe_capture_id; 0: not capturing)."""
    cid = C.c_uint64()
    check(load_library().fe_capture_id(stream, C.byref(cid)))
    return int(cid.value)


def graph_retired(cid: int) -> int:
    """The graph captured under *cid* and its executables are gone: its launches' counter groups may serve others
    (fe_graph_retired); returns how many groups came back."""
    rc = int(load_library().fe_graph_retired(int(cid)))
    if rc < 0:
        check(rc)
    return rc


def tail_plant(stream: int, value: int) -> None:
    """Test hook (fe_tail_plant): leave a stale ticket in the counter group of *stream*."""
    check(load_library().fe_tail_plant(stream, int(value)))


_LAST_LAUNCH = ("valid", "dynamic_walk", "temporal_loads", "write_through_stores", "blocks", "waves_per_block", "kind", "bodies", "tiles",
                "static_tiles")


def last_launch_info() -> dict:
    """What the launcher decided for the MFMA launch this thread enqueued last (fe_last_launch_info): the walk, the cache hints
    of loads and stores, the grid.  ``{}`` before the first such launch."""
    buf = (C.c_int64 * len(_LAST_LAUNCH))()
    n = int(load_library().fe_last_launch_info(buf, len(_LAST_LAUNCH)))
    if n < 0:
        check(n)
    d = {k: int(buf[i]) for i, k in enumerate(_LAST_LAUNCH[:n])}
    if not d.get("valid"):
        return {}
    d["interleaved"] = bool(d["kind"] & 4)
    d["quarter_tail"] = bool(d["kind"] & 8)
    d["staggered_start"] = bool(d["kind"] & 16)
    d["kind"] = "B build interleaved" if d["kind"] & 4 else "default"
    return d


def set_phase_priority_p5(on: bool) -> bool:
    """Phase priorities in the eight-wave p = 5 kernels (fe_set_phase_priority_p5); returns the previous setting."""
    return bool(load_library().fe_set_phase_priority_p5(1 if on else 0))


def set_div_quarter_tail(on: bool) -> bool:
    """The ragged last round of short interleaved div launches as quarter tiles (fe_set_div_quarter_tail); returns the previous
    setting."""
    return bool(load_library().fe_set_div_quarter_tail(1 if on else 0))


def set_grad_quarter_tail(on: bool) -> bool:
    """The ragged last round of short grad launches (one field, static walk) as quarter tiles (fe_set_grad_quarter_tail); returns
    the previous setting."""
    return bool(load_library().fe_set_grad_quarter_tail(int(on)))


def set_grad_staggered_start(on: bool) -> bool:
    """Short grad launches (one field, static walk, 2.5 to 4.5 rounds) start every second CU of an XCD half a tile period late
    (fe_set_grad_staggered_start); returns the previous setting."""
    return bool(load_library().fe_set_grad_staggered_start(1 if on else 0))


def set_div_interleave(tiles: int) -> int:
    """Short div launches of at most *tiles* tiles run on the interleaved kernel (fe_set_div_interleave; 0 = never); returns the
    previous setting."""
    return int(load_library().fe_set_div_interleave(int(tiles)))


def set_write_through_mib(mib: int) -> int:
    """grad launches that write at most *mib* MiB store write-through instead of non-temporally (fe_set_write_through_mib;
    0 = never); returns the previous setting.  A tuning knob."""
    return int(load_library().fe_set_write_through_mib(int(mib)))


def set_temporal_loads_mib(mib: int) -> int:
    """Launches whose inputs are at most *mib* MiB fetch their streamed operand with plain (cacheable) loads instead of
    non-temporal ones (fe_set_temporal_loads_mib; 0 = never); returns the previous setting.  A tuning knob."""
    return int(load_library().fe_set_temporal_loads_mib(int(mib)))
