"""
Can torch use memory that was mapped with HIP's virtual-memory API (hipMemCreate / hipMemMap)?

The split allocator (feinsum_amd/placement.py) hands out such memory; callers hold it as torch tensors.  Tried here, on a
64 MiB range made of 2 MiB handles: hipPointerGetAttributes, torch.as_tensor over __cuda_array_interface__ (with and
without an explicit device), torch.from_dlpack over a hand-made DLPack capsule; then torch kernels on the tensor (fill,
sum, copy to host) and a feinsum launch writing into it.

    python tools/vmm_torch_probe.py
"""

from __future__ import annotations

import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
MIB = 1 << 20


class Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class Prop(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("handle_type", ctypes.c_int), ("location", Loc), ("win32", ctypes.c_void_p),
                ("compression", ctypes.c_ubyte), ("rdma", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class Access(ctypes.Structure):
    _fields_ = [("location", Loc), ("flags", ctypes.c_int)]


def ck(rc, what):
    if rc != 0:
        hip.hipGetErrorString.restype = ctypes.c_char_p
        raise RuntimeError(f"{what}: {hip.hipGetErrorString(rc).decode()}")


def vmm_range(nbytes: int, piece: int = 2 * MIB):
    prop = Prop(type=1, handle_type=0, location=Loc(1, 0))
    va = ctypes.c_void_p()
    ck(hip.hipMemAddressReserve(ctypes.byref(va), ctypes.c_size_t(nbytes), ctypes.c_size_t(piece), None, ctypes.c_ulonglong(0)),
       "hipMemAddressReserve")
    handles = []
    for off in range(0, nbytes, piece):
        h = ctypes.c_void_p()
        ck(hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(piece), ctypes.byref(prop), ctypes.c_ulonglong(0)), "hipMemCreate")
        ck(hip.hipMemMap(ctypes.c_void_p(va.value + off), ctypes.c_size_t(piece), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)),
           "hipMemMap")
        handles.append(h)
    acc = Access(Loc(1, 0), 3)
    ck(hip.hipMemSetAccess(va, ctypes.c_size_t(nbytes), ctypes.byref(acc), ctypes.c_size_t(1)), "hipMemSetAccess")
    return va.value, handles


class Holder:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
    n = 64 * MIB
    ptr, _handles = vmm_range(n)
    print(f"VMM range at {ptr:#x}, {n // MIB} MiB of 2 MiB handles")
    shape = (n // 8,)
    t = None
    for label, make in (
        ("as_tensor(__cuda_array_interface__, device='cuda:0')", lambda: torch.as_tensor(Holder(ptr, shape), device="cuda:0")),
        ("as_tensor(__cuda_array_interface__)", lambda: torch.as_tensor(Holder(ptr, shape))),
    ):
        try:
            cand = make()
            print(f"{label}: OK  data_ptr {cand.data_ptr():#x} (zero copy: {cand.data_ptr() == ptr}) device {cand.device}")
            if t is None and cand.data_ptr() == ptr:
                t = cand
        except Exception as exc:  # noqa: BLE001
            print(f"{label}: FAILED {type(exc).__name__}: {exc}")
    if t is None:
        print("no zero-copy route worked")
        return 1
    t.fill_(1.5)
    torch.cuda.synchronize()
    print("fill_ + sum:", float(t.sum().item()), "expected", 1.5 * shape[0])
    v = t[: 3 * 1000 * 35].view(3, 1000, 35)
    import feinsum_amd as f
    from feinsum_amd.measure import generate_host_input_arrays

    expr = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E")), f.array("R", (3, 35, 35)), f.array("u", ("E", 35)))
    host = generate_host_input_arrays(expr, 1000)
    dev = {k: torch.from_numpy(a).to("cuda:0") for k, a in host.items()}
    ref = f.evaluate(expr, 0, dev, wait=True)["_fe_out"]
    out = f.evaluate(expr, 0, dev, out_dict={"_fe_out": v}, wait=True)["_fe_out"]
    print("feinsum launch into the VMM tensor: same object", out.data_ptr() == v.data_ptr(), " equal to a plain launch",
          bool(torch.equal(out, ref)))
    print("host copy:", float(v.cpu().abs().max()))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
