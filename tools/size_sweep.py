#!/usr/bin/env python
"""
The p = 4 launches over the element count, as shipped (outputs from the split allocator, the launcher's own choices):

    python tools/size_sweep.py [grad div facemass graddiv pipeline] [E ...]

Per line: kernel time of back-to-back launches (median of five batches, HIP events), fraction of the HBM roofline, and what the
launcher chose (fe_last_launch_info).
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure  # noqa: E402

ALL = ("grad", "div", "facemass", "graddiv", "pipeline")
names = [a for a in sys.argv[1:] if a in ALL] or list(ALL)
sizes = [int(float(a)) for a in sys.argv[1:] if a not in ALL] or [10_000, 20_000, 50_000, 80_000, 98_304, 100_000, 100_007, 131_072, 200_000,
                                                                   300_000, 500_000, 1_000_000, 2_000_000]


def stages_of(what, E):
    g = torch.Generator(device="cuda").manual_seed(3)

    def inputs(expr):
        return {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64,
                              device="cuda", generator=g) for n in sorted(expr.all_args)}
    grad, div, fm = dg.grad(), dg.div(), dg.face_mass(4)
    gd = inputs(grad)
    dd = dict(inputs(div), J=gd["J"], R=gd["R"])
    return {"grad": [(grad, gd)], "div": [(div, dd)], "facemass": [(fm, inputs(fm))], "graddiv": [(div, dd), (grad, gd)],
            "pipeline": [(div, dd), (grad, gd), (fm, inputs(fm))]}[what]


for what in names:
    for E in sizes:
        stages = stages_of(what, E)
        nbytes = sum(measure._get_footprint_gbytes(e, E) * 1e9 for e, _ in stages) - (8.0 * (9 * E + 3 * 35 * 35) if len(stages) > 1 else 0.0)
        outs = [measure.generate_out_arrays(0, e, E, split=True) for e, _ in stages]
        op = f.bind_operator(stages, 0, out_dicts=outs)
        n = max(20, min(400, int(4e7 / E)))
        op.time_batch(5 * n)
        t = sorted(op.time_batch(n) / n for _ in range(5))[2]
        info = _hip.last_launch_info()
        chose = ", ".join(k for k, on in (("tickets", info.get("dynamic_walk")), ("plain loads", info.get("temporal_loads")),
                                          ("write-through stores", info.get("write_through_stores")), ("quarter tiles", info.get("quarter_tail")), ("staggered start", info.get("staggered_start")),
                                          ("interleaved build", info.get("interleaved"))) if on) or "static walk, non-temporal"
        print(f"{what:8s} E={E:8d}: {t * 1e6:8.2f} us = {nbytes / t / 8e12:.3f} of the roofline   [{chose}]", flush=True)
        del op, outs, stages
