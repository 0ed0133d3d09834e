#!/usr/bin/env python
"""
One fused launch against the same stages as separate launches, back to back on the same arrays, in one process:

    python tools/fuse_ab.py [graddiv pipeline] [E ...]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import measure  # noqa: E402

names = [a for a in sys.argv[1:] if a in ("graddiv", "pipeline")] or ["graddiv", "pipeline"]
sizes = [int(float(a)) for a in sys.argv[1:] if a not in ("graddiv", "pipeline")] or [20_000, 50_000, 98_304, 100_000, 131_072, 200_000, 400_000, 1_000_000]


def stages_of(what, E):
    g = torch.Generator(device="cuda").manual_seed(3)

    def inputs(expr):
        return {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64,
                              device="cuda", generator=g) for n in sorted(expr.all_args)}
    grad, div, fm = dg.grad(), dg.div(), dg.face_mass(4)
    gd = inputs(grad)
    dd = dict(inputs(div), J=gd["J"], R=gd["R"])
    return [(div, dd), (grad, gd)] + ([(fm, inputs(fm))] if what == "pipeline" else [])


def timed(op, n):
    op.time_batch(10)
    return sorted(op.time_batch(n) / n for _ in range(5))[2]


for what in names:
    for E in sizes:
        stages = stages_of(what, E)
        nbytes = sum(measure._get_footprint_gbytes(e, E) * 1e9 for e, _ in stages) - 8.0 * (9 * E + 3 * 35 * 35)
        outs = [measure.generate_out_arrays(0, e, E, split=True) for e, _ in stages]
        fused = f.bind_operator(stages, 0, out_dicts=outs)
        apart = f.bind_operator(stages, 0, out_dicts=outs, fuse=False)
        n = max(20, min(400, int(4e7 / E)))
        timed(fused, 5 * n)
        best = {0: 1e9, 1: 1e9}
        for rep in range(3):
            for mode, op in ((0, fused), (1, apart)):
                best[mode] = min(best[mode], timed(op, n))
        a, b = best[0], best[1]
        print(f"{what:8s} E={E:8d}: one launch {a * 1e6:7.2f} us = {nbytes / a / 8e12:.3f}   {len(stages)} launches {b * 1e6:7.2f} us = {nbytes / b / 8e12:.3f}   "
              f"({(b / a - 1) * 100:+.1f} %)   entry points {fused.entry_points}", flush=True)
        del fused, apart, outs, stages
