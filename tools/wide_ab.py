#!/usr/bin/env python
"""
A/B in one process on the SAME arrays: p = 4 grad / div on the default kernels (two waves per SIMD, operator fragments in
registers) against the sixteen-waves-per-CU kernels (fe_set_wide_blocks: fragments in LDS, two eight-wave blocks per CU).

    python tools/wide_ab.py [grad div] [E ...]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure  # noqa: E402

CASES = {"grad": dg.grad, "div": dg.div}
names = [a for a in sys.argv[1:] if a in CASES] or ["grad", "div"]
sizes = [int(float(a)) for a in sys.argv[1:] if a not in CASES] or [20_000, 50_000, 98_304, 100_000, 131_072, 200_000, 500_000, 1_000_000]


def timed(bound, q, n):
    bound.time_batch(10, q.stream_ptr)
    return sorted(bound.time_batch(n, q.stream_ptr) / n for _ in range(5))[2]


for what in names:
    expr = CASES[what]()
    for E in sizes:
        nbytes = measure._get_footprint_gbytes(expr, E) * 1e9
        host = measure.generate_host_input_arrays(expr, E, np_seed=0)
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        outs = measure.generate_out_arrays(0, expr, E, split=True)
        q, bound, _ = measure._bind(expr, 0, dev, outs, None)
        n = max(20, min(400, int(4e7 / E)))
        _hip.set_wide_blocks(0)
        timed(bound, q, 5 * n)   # settle
        best = {0: 1e9, 1: 1e9}
        for rep in range(3):
            for mode in (0, 1):
                _hip.set_wide_blocks(mode)
                best[mode] = min(best[mode], timed(bound, q, n))
        _hip.set_wide_blocks(0)
        f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
        ref = {k: v.clone() for k, v in outs.items()}
        for v in outs.values():
            v.zero_()
        _hip.set_wide_blocks(1)
        f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
        _hip.set_wide_blocks(0)
        err = max(float(((outs[k] - ref[k]).abs().max() / ref[k].abs().max()).item()) for k in ref)
        a, b = best[0], best[1]
        print(f"{what} E={E:8d}: default {a * 1e6:7.2f} us = {nbytes / a / 8e12:.3f}   wide {b * 1e6:7.2f} us = {nbytes / b / 8e12:.3f}   "
              f"({(b / a - 1) * 100:+.1f} %)   max rel diff {err:.1e}", flush=True)
        del dev, outs, bound, ref
