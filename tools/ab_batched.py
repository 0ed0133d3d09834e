"""A/B of library builds on the batched grad / div launches (fe_time_launches through ctypes).

    python tools/ab_batched.py grad|div b E rounds lib1.so lib2.so ...
"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from feinsum_amd._hip import ArgPack  # noqa: E402

fam, b, E, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
libs = sys.argv[5:]
Np = 35
g = torch.Generator(device="cuda").manual_seed(0)
J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
D = torch.rand((3, Np, Np), dtype=torch.float64, device="cuda", generator=g)
ushape, oshape = ((E, Np), (3, E, Np)) if fam == "grad" else ((3, E, Np), (E, Np))
us = [torch.rand(ushape, dtype=torch.float64, device="cuda", generator=g) for _ in range(b)]
outs = [torch.empty(oshape, dtype=torch.float64, device="cuda") for _ in range(b)]
va = (C.c_void_p * b)(*[t.data_ptr() for t in us])
oa = (C.c_void_p * b)(*[t.data_ptr() for t in outs])
pack = ArgPack()
pack.J, pack.D, pack.u, pack.out = J.data_ptr(), D.data_ptr(), us[0].data_ptr(), outs[0].data_ptr()
pack.v, pack.outs = va, oa
pack.E, pack.Np, pack.b, pack.ndim = E, Np, b, 3
family = 1 if fam == "grad" else 2
fns = []
for path in libs:
    lib = C.CDLL(path)
    lib.fe_time_launches.argtypes = [C.c_int32, C.POINTER(ArgPack), C.c_int32, C.c_void_p, C.POINTER(C.c_float)]
    fns.append(lib.fe_time_launches)
ms = C.c_float()
times = [[] for _ in libs]
for fn in fns:
    assert fn(family, C.byref(pack), 3, None, C.byref(ms)) == 0
for _ in range(rounds):
    for k, fn in enumerate(fns):
        assert fn(family, C.byref(pack), 10, None, C.byref(ms)) == 0
        times[k].append(ms.value / 10)
for path, t in zip(libs, times):
    t.sort()
    print(f"{fam} b={b} E={E} {path:34s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}")
