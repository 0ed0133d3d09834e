#!/usr/bin/env python
"""
A/B in one process on the SAME arrays: the eight-wave p = 5 kernels (grad, div) with and without phase priorities
(fe_set_phase_priority_p5).     python tools/p5_phase_ab.py [E ...]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure  # noqa: E402

sizes = [int(float(a)) for a in sys.argv[1:]] or [200_000, 1_000_000, 2_000_000]


def timed(bound, q, n):
    bound.time_batch(10, q.stream_ptr)
    return sorted(bound.time_batch(n, q.stream_ptr) / n for _ in range(5))[2]


for what, expr in (("grad p5", dg.grad(56)), ("div p5", dg.div(56))):
    for E in sizes:
        flops = f.count_ops(expr, long_dim_length=E)
        host = measure.generate_host_input_arrays(expr, E, np_seed=0)
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        outs = measure.generate_out_arrays(0, expr, E, split=True)
        q, bound, _ = measure._bind(expr, 0, dev, outs, None)
        n = max(20, min(200, int(2e7 / E)))
        _hip.set_phase_priority_p5(False)
        timed(bound, q, 3 * n)
        best = {0: 1e9, 1: 1e9}
        for rep in range(3):
            for mode in (0, 1):
                _hip.set_phase_priority_p5(bool(mode))
                best[mode] = min(best[mode], timed(bound, q, n))
        _hip.set_phase_priority_p5(False)
        f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
        ref = {k: v.clone() for k, v in outs.items()}
        _hip.set_phase_priority_p5(True)
        f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
        _hip.set_phase_priority_p5(False)
        same = all(torch.equal(outs[k], ref[k]) for k in ref)
        a, b = best[0], best[1]
        print(f"{what} E={E:8d}: default {a * 1e3:7.4f} ms = {flops / a * 1e-12:5.2f} TFLOP/s   phase priorities {b * 1e3:7.4f} ms = {flops / b * 1e-12:5.2f} TFLOP/s   "
              f"({(b / a - 1) * 100:+.1f} %)   same bits {same}", flush=True)
        del dev, outs, bound, ref
