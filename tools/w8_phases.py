#!/usr/bin/env python
"""
Per-wave phase times of the eight-wave p = 5 grad kernel (experiment build: bash tools/build_experiments.sh).

    python tools/w8_phases.py [E]

Loads build/libfeinsum_hip_exp.so directly (never the package's library), launches fe_grad3d_f64 at Np = 56 and reads
fe_dbg_w8: shader cycles spent waiting for the tile / in the MFMA phase / in the epilogue, tiles done, hardware slot.
"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Np = 56
lib = ctypes.CDLL(str(ROOT / "build" / "libfeinsum_hip_exp.so"))
lib.fe_last_error.restype = ctypes.c_char_p
g = torch.Generator(device="cuda").manual_seed(0)
J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
D = torch.rand((3, Np, Np), dtype=torch.float64, device="cuda", generator=g)
u = torch.rand((E, Np), dtype=torch.float64, device="cuda", generator=g)
out = torch.zeros((3, E, Np), dtype=torch.float64, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731


def launch(variant=0):
    rc = lib.fe_grad3d_f64(p(J), p(D), p(u), p(out), ctypes.c_int64(E), ctypes.c_int32(Np), ctypes.c_int32(variant), ctypes.c_void_p(0))
    assert rc == 0, lib.fe_last_error()


flops = 2.0 * 3 * Np * Np * E + 2.0 * 9 * Np * E
names = {0: "product kernel", 1016: "next unit requested before the last stores", 1064: "priority + LDS tile tickets", 1080: "both", 1001: "no MFMAs", 1002: "no stores", 1008: "no loads", 1003: "no MFMAs, no stores",
         1009: "no MFMAs, no loads", 1010: "no stores, no loads", 1011: "neither (LDS + VALU skeleton)", 1032: "with per-wave stamps"}
for rnd in range(3):
    for v, what in names.items():
        for _ in range(10):
            launch(v)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            launch(v)
        t1.record()
        t1.synchronize()
        ms = t0.elapsed_time(t1) / 20
        print(f"grad Np = 56, E = {E}, {what:32s}: {ms:.4f} ms ({flops / ms * 1e-9:.1f} TFLOP/s counted)", flush=True)
launch(1032)
torch.cuda.synchronize()
n = 2048
buf = (ctypes.c_ulonglong * (n * 8))()
assert lib.fe_dbg_read_w8(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
a = a[a[:, 3] > 0]
hw = a[:, 4]
slot, simd, cu = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15
dur = (a[:, 6] - a[:, 5]) / 100.0      # us
tmin = a[:, 5].min()
print(f"waves with tiles: {len(a)}; loop durations us: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; "
      f"last end {(a[:, 6].max() - tmin) / 100.0:.1f} us")
for s in sorted(set(slot.tolist())):
    m = slot == s
    w, mf, ep, nt = a[m, 0], a[m, 1], a[m, 2], a[m, 3]
    print(f"slot {s}: {m.sum():4d} waves  tiles/wave {nt.mean():6.2f}  per tile: wait {np.mean(w / nt):7.0f}  MFMA phase {np.mean(mf / nt):7.0f}  "
          f"epilogue {np.mean(ep / nt):7.0f} cycles   loop {dur[m].mean():.1f} us (ends {((a[m, 6] - tmin) / 100.0).mean():.1f})")
print("SIMD x slot occupancy of the first CU seen:", sorted(zip(simd[cu == cu[0]][:8].tolist(), slot[cu == cu[0]][:8].tolist())))
