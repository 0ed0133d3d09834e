#!/usr/bin/env python
"""
A/B in one process (experiment build): the body orders of the fused div + grad launch (0: every block div then grad, 1: the
younger half of the grid grad first, 2: the odd blocks grad first) with the static walk and with the dynamic walk
(fe_set_tail_rounds).  Outputs in arrays of the split allocator (feinsum_amd.placement) and in torch allocations.

    bash tools/build_experiments.sh && python tools/fused_order_dyn_ab.py
"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from feinsum_amd import placement  # noqa: E402

E, Np = 1_000_000, 35
lib = ctypes.CDLL(str(ROOT / "build" / "libfeinsum_hip_exp.so"))
lib.fe_last_error.restype = ctypes.c_char_p
g = torch.Generator(device="cuda").manual_seed(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731


def timed(J, D, u, v, go, do, order, n=40):
    os.environ["FE_FUSED_ORDER"] = str(order)

    def launch():
        rc = lib.fe_graddiv3d_f64(P(J), P(D), P(u), P(v), P(go), P(do), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(0), ctypes.c_void_p(0))
        assert rc == 0, lib.fe_last_error()
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            launch()
        t1.record()
        t1.synchronize()
        ts.append(t0.elapsed_time(t1) / n)
    return sorted(ts)[2]


def rnd(shape):
    return torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)


J, D, u, v = rnd((3, 3, E)), rnd((3, Np, Np)), rnd((E, Np)), rnd((3, E, Np))
for what, alloc in (("split allocator", lambda s: placement.zeros(s, torch.float64, "cuda:0")),
                    ("torch allocations", lambda s: torch.zeros(s, dtype=torch.float64, device="cuda"))):
    go, do = alloc((3, E, Np)), alloc((E, Np))
    timed(J, D, u, v, go, do, 1, n=200)       # settle
    for rep in range(3):
        for rounds in (-1, 1 << 20):
            lib.fe_set_tail_rounds(ctypes.c_int32(rounds))
            cells = []
            for o in (0, 1, 2):
                t = timed(J, D, u, v, go, do, o)
                cells.append(f"order {o}: {t:.4f} ms ({2312.0294e6 / t / 8e9 * 100:.1f} %)")
            print(f"{what}, {'static ' if rounds < 0 else 'dynamic'} walk: " + "  ".join(cells), flush=True)
