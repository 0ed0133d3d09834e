#!/usr/bin/env python
"""grad p = 4 at E = 1e5 (the reference's default long_dim_length): does the placement matter at this size?  A class boundary
is located with the E = 1e6 launch; the 84 MB output of the E = 1e5 launch is then moved across it in 4 MiB steps."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from feinsum_amd import _hip  # noqa: E402

lib = _hip.load_library()
Np = 35
g = torch.Generator(device="cuda").manual_seed(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
MIB, GIB = 1 << 20, 1 << 30
arena = torch.empty(66 * GIB, dtype=torch.uint8, device="cuda")


def setup(E):
    J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
    D = torch.rand((3, Np, Np), dtype=torch.float64, device="cuda", generator=g)
    u = torch.rand((E, Np), dtype=torch.float64, device="cuda", generator=g)
    return J, D, u


def timed(ops, out, E, n=20):
    J, D, u = ops

    def launch():
        rc = lib.fe_grad3d_f64(P(J), P(D), P(u), P(out), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(0), ctypes.c_void_p(0))
        assert rc == 0
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


view = lambda base, E: arena[base:base + 3 * E * Np * 8].view(torch.float64).view(3, E, Np)   # noqa: E731
E6 = 1_000_000
ops6 = setup(E6)
times = {b: timed(ops6, view(b, E6), E6, n=6) for b in range(0, 64 * GIB, 128 * MIB)}
best = min(times, key=times.get)
fine = {b: timed(ops6, view(b, E6), E6, n=10) for b in range(max(0, best - 256 * MIB), best + 256 * MIB, 32 * MIB)}
lo = min(fine.values())
plateau = [b for b, t in fine.items() if t < lo * 1.01]
boundary = (min(plateau) + max(plateau)) // 2 + 3 * E6 * Np * 8 // 2
print(f"E = 1e6: scan median {sorted(times.values())[len(times) // 2]:.4f}, best {lo:.4f} ms; class boundary near {boundary / GIB:.3f} GiB")
E5 = 100_000
ops5 = setup(E5)
plane = E5 * Np * 8
rows = []
for off in range(-160 * MIB, 64 * MIB, 4 * MIB):
    base = (boundary + off) // (2 * MIB) * (2 * MIB)
    rows.append((off, timed(ops5, view(base, E5), E5, n=50)))
ts = sorted(t for _, t in rows)
print(f"E = 1e5: output start relative to the boundary (MiB) -> us: " + " ".join(f"{off // MIB}:{t * 1e3:.1f}" for off, t in rows))
print(f"E = 1e5: median {ts[len(ts) // 2] * 1e3:.2f} us, min {ts[0] * 1e3:.2f} us, max {ts[-1] * 1e3:.2f} us (roofline fraction at the minimum: "
      f"{(1192.0 * E5 + 29400) / (ts[0] * 1e-3) / 8e12:.3f})")
far = timed(ops5, view(8 * GIB, E5), E5, n=50)
print(f"E = 1e5 far from any boundary (8 GiB): {far * 1e3:.2f} us")
