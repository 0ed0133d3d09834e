#!/usr/bin/env python
"""
grad p = 4 at E = 1e6 in a FAST placement (output across a class boundary, found by scanning an arena): the product
kernel against the experiment build's "no MFMAs" (1001) and "no stores" (1002) variants, in one process.

    python tools/grad_fast_decomposition.py
"""
import ctypes
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
E, Np = 1_000_000, 35
lib = ctypes.CDLL(str(ROOT / "build" / "libfeinsum_hip_exp.so"))
lib.fe_last_error.restype = ctypes.c_char_p
g = torch.Generator(device="cuda").manual_seed(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
MIB, GIB = 1 << 20, 1 << 30
J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
D = torch.rand((3, Np, Np), dtype=torch.float64, device="cuda", generator=g)
u = torch.rand((E, Np), dtype=torch.float64, device="cuda", generator=g)
arena = torch.empty(66 * GIB, dtype=torch.uint8, device="cuda")
nb = 3 * E * Np * 8


def timed(out, variant, n=20):
    def launch():
        rc = lib.fe_grad3d_f64(P(J), P(D), P(u), P(out), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(variant), ctypes.c_void_p(0))
        assert rc == 0, lib.fe_last_error()
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        launch()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / n


view = lambda base: arena[base:base + nb].view(torch.float64).view(3, E, Np)   # noqa: E731
times = {b: timed(view(b), 0, n=6) for b in range(0, 64 * GIB, 128 * MIB)}
srt = sorted(times.values())
best = min(times, key=times.get)
slow = min(times, key=lambda b: abs(times[b] - srt[len(srt) // 2]))
print(f"scan: median {srt[len(srt) // 2]:.4f} min {srt[0]:.4f} ms; fast base {best // MIB} MiB, a median base {slow // MIB} MiB")
for what, base in (("fast placement", best), ("median placement", slow)):
    for rnd in range(2):
        print(f"{what}: " + "  ".join(f"{name} {timed(view(base), v):.4f} ms" for name, v in
                                      (("product", 0), ("no MFMAs", 1001), ("no stores", 1002), ("no priority balancing", 1064))), flush=True)
