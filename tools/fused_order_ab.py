#!/usr/bin/env python
"""
A/B in one process (experiment build): the fused div + grad launch with every block running div then grad (0), with
the younger half of the grid running grad first (1), with the odd blocks running grad first (2).  Arrays in one arena
across a class boundary found by scanning (as bench.py does), and with one allocation per array.

    python tools/fused_order_ab.py
"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
E, Np = 1_000_000, 35
lib = ctypes.CDLL(str(ROOT / "build" / "libfeinsum_hip_exp.so"))
lib.fe_last_error.restype = ctypes.c_char_p
g = torch.Generator(device="cuda").manual_seed(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731


def timed(J, D, u, v, go, do, order, n=20):
    os.environ["FE_FUSED_ORDER"] = str(order)

    def launch():
        rc = lib.fe_graddiv3d_f64(P(J), P(D), P(u), P(v), P(go), P(do), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(0), ctypes.c_void_p(0))
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


def rnd(shape):
    return torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)


# (a) one allocation per array
J, D, u, v = rnd((3, 3, E)), rnd((3, Np, Np)), rnd((E, Np)), rnd((3, E, Np))
go, do = torch.zeros((3, E, Np), dtype=torch.float64, device="cuda"), torch.zeros((E, Np), dtype=torch.float64, device="cuda")
for rnd_i in range(3):
    print("separate allocations: " + "  ".join(f"order {o}: {timed(J, D, u, v, go, do, o):.4f} ms" for o in (0, 1, 2)), flush=True)
ref_go, ref_do = go.clone(), do.clone()
timed(J, D, u, v, go, do, 1, n=1)
assert torch.equal(go, ref_go) and torch.equal(do, ref_do)
timed(J, D, u, v, go, do, 2, n=1)
assert torch.equal(go, ref_go) and torch.equal(do, ref_do)
print("results identical for the three orders")

# (b) outputs in an arena, moved through it: best position per order
MIB, GIB = 1 << 20, 1 << 30
arena = torch.empty(66 * GIB, dtype=torch.uint8, device="cuda")
nb_go, nb_do = 3 * E * Np * 8, E * Np * 8


def views(base):
    d = arena[base:base + nb_do].view(torch.float64).view(E, Np)
    off = (base + nb_do + 64 * MIB + 2 * MIB - 1) // (2 * MIB) * (2 * MIB)
    gg = arena[off:off + nb_go].view(torch.float64).view(3, E, Np)
    return gg, d


best = {}
for o in (0, 1, 2):
    times = {}
    for base in range(0, 64 * GIB, 256 * MIB):
        gg, d = views(base)
        times[base] = timed(J, D, u, v, gg, d, o, n=6)
    b = min(times, key=times.get)
    gg, d = views(b)
    fine = sorted(timed(J, D, u, v, gg, d, o) for _ in range(3))[1]
    srt = sorted(times.values())
    best[o] = fine
    print(f"order {o}: scan median {srt[len(srt) // 2]:.4f} min {srt[0]:.4f}; at the best position (base {b // MIB} MiB): {fine:.4f} ms", flush=True)

# (c) the three-body launch: div + grad + face-mass x 4; written arrays in split order (placement.split_order) in the arena
nf, Nfp, nb = 4, 15, 4
Jf, R = rnd((E, nf)), rnd((nf, Np, Nfp))
fv = [rnd((nf, E, Nfp)) for _ in range(nb)]
PtrArr = ctypes.c_void_p * nb
fptr = PtrArr(*[t.data_ptr() for t in fv])


def timed3(gg, d, lifts, order, n=20):
    os.environ["FE_FUSED_ORDER"] = str(order)
    lptr = PtrArr(*[t.data_ptr() for t in lifts])

    def launch():
        rc = lib.fe_waveop3d_f64(P(J), P(D), P(u), P(gg), P(v), P(d), P(Jf), P(R), fptr, lptr, ctypes.c_int64(E), ctypes.c_int32(Np),
                                 ctypes.c_int32(nf), ctypes.c_int32(Nfp), ctypes.c_int32(nb), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_void_p(0))
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


def views3(base):
    off = base
    outs = []
    for nbytes, shape in ((nb_do, (E, Np)), (nb_do, (E, Np)), (nb_do, (E, Np)), (nb_go, (3, E, Np)), (nb_do, (E, Np)), (nb_do, (E, Np))):
        off = (off + 2 * MIB - 1) // (2 * MIB) * (2 * MIB)
        outs.append(arena[off:off + nbytes].view(torch.float64).view(shape))
        off += nbytes + 64 * MIB
    d, l0, l1, gg, l2, l3 = outs          # split order: div, lift0, lift1 | grad | lift2, lift3
    return gg, d, [l0, l1, l2, l3]


lifts_sep = [torch.zeros((E, Np), dtype=torch.float64, device="cuda") for _ in range(nb)]
for rnd_i in range(2):
    print("three bodies, separate allocations: " + "  ".join(f"order {o}: {timed3(go, do, lifts_sep, o):.4f} ms" for o in (0, 1, 2, 3)), flush=True)
ref = [t.clone() for t in [go, do] + lifts_sep]
for o in (1, 2, 3):
    timed3(go, do, lifts_sep, o, n=1)
    assert all(torch.equal(a, b) for a, b in zip(ref, [go, do] + lifts_sep))
print("three bodies: results identical for the three orders")
for o in (0, 3, 0, 3):
    times = {}
    for base in range(0, 62 * GIB, 256 * MIB):
        times[base] = timed3(*views3(base), o, n=4)
    b = min(times, key=times.get)
    fine = sorted(timed3(*views3(b), o) for _ in range(3))[1]
    srt = sorted(times.values())
    print(f"three bodies, order {o}: scan median {srt[len(srt) // 2]:.4f} min {srt[0]:.4f}; at the best position (base {b // MIB} MiB): {fine:.4f} ms", flush=True)
