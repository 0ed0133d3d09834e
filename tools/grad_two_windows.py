#!/usr/bin/env python
"""grad p = 4, E = 1e6 (experiment build): product walk against the two-window walk (variant 1128), output moved through an arena."""
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
ref = torch.zeros((3, E, Np), dtype=torch.float64, device="cuda")
timed(ref, 0, n=1)
chk = torch.zeros((3, E, Np), dtype=torch.float64, device="cuda")
timed(chk, 1128, n=1)
print("two-window walk: results identical:", torch.equal(ref, chk))
for rnd in range(2):
    for v in (0, 1128):
        times = {b: timed(view(b), v, n=6) for b in range(0, 64 * GIB, 64 * MIB)}
        srt = sorted(times.values())
        best = min(times, key=times.get)
        fine = sorted(timed(view(best), v) for _ in range(3))
        print(f"variant {v}: scan median {srt[len(srt) // 2]:.4f} min {srt[0]:.4f} max {srt[-1]:.4f} ms; best base {best // MIB} MiB: {fine[1]:.4f} ms "
              f"(positions within 1 % of the minimum: {sum(t < 1.01 * srt[0] for t in srt)})", flush=True)
