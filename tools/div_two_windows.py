#!/usr/bin/env python
"""
div p = 4 at E = 1e6 (experiment build): the product walk against the two-window walk (variant 1004: even steps in
the first half of the elements, odd steps in the second) with the OUTPUT moved through an arena -- a class boundary
in the middle of the output then splits the two write windows.

    python tools/div_two_windows.py
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
u = torch.rand((3, E, Np), dtype=torch.float64, device="cuda", generator=g)
arena = torch.empty(66 * GIB, dtype=torch.uint8, device="cuda")
nb = E * Np * 8


def timed(out, variant, n=20):
    def launch():
        rc = lib.fe_div3d_f64(P(J), P(D), P(u), P(out), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(variant), ctypes.c_void_p(0))
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


view = lambda base: arena[base:base + nb].view(torch.float64).view(E, Np)   # noqa: E731
ref = torch.zeros((E, Np), dtype=torch.float64, device="cuda")
timed(ref, 0, n=1)
chk = torch.zeros((E, Np), dtype=torch.float64, device="cuda")
timed(chk, 1004, n=1)
print("two-window walk: results identical:", torch.equal(ref, chk))
for v in (0, 1004):
    times = {b: timed(view(b), v, n=6) for b in range(0, 64 * GIB, 64 * MIB)}
    srt = sorted(times.values())
    best = min(times, key=times.get)
    fine = sorted(timed(view(best), v) for _ in range(3))
    print(f"variant {v}: scan median {srt[len(srt) // 2]:.4f} min {srt[0]:.4f} max {srt[-1]:.4f} ms; best base {best // MIB} MiB: {fine[1]:.4f} ms "
          f"(positions within 1 % of the minimum: {sum(t < 1.01 * srt[0] for t in srt)})", flush=True)
    fast = sorted(b // MIB for b, t in times.items() if t < 0.985 * srt[len(srt) // 2])[:12]
    print("   fast bases (MiB):", fast)
