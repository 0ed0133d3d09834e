#!/usr/bin/env python
"""
A/B in one process on the SAME arrays of a launcher knob against its off state (round 5's knob=phase -- phase priorities on the
p = 4 kernels -- was measured with this tool and removed: profiles/r05/phase_priorities_ab.txt).

    python tools/knob_ab.py [grad div graddiv pipeline] [E ...] [knob=ilv|quarter|gquarter|gstagger|tickets3|tickets2] [ilv=on]

knob=ilv: div launches on the kernel whose B build is interleaved into the matrix phase (fe_set_div_interleave) instead;
knob=tickets3 / tickets2: the dynamic walk from three / two full rounds on (fe_set_tail_min_rounds); ilv=on: with the interleaved div.
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure  # noqa: E402

ALL = ("grad", "div", "graddiv", "pipeline")
knob = ([a.split("=")[1] for a in sys.argv[1:] if a.startswith("knob=")] or ["ilv"])[0]
setter = {"ilv": _hip.set_div_interleave, "quarter": lambda v: _hip.set_div_quarter_tail(bool(v)), "gquarter": lambda v: _hip.set_grad_quarter_tail(bool(v)), "gquarter4": lambda v: _hip.set_grad_quarter_tail(4 if v else 1), "gstagger": lambda v: _hip.set_grad_staggered_start(bool(v)),
          "tickets3": lambda v: _hip.set_tail_min_rounds(3 if v else 4), "tickets2": lambda v: _hip.set_tail_min_rounds(2 if v else 4)}[knob]
label = {"ilv": "interleaved B build", "quarter": "quarter-tile tail", "gquarter": "quarter-tile tail (grad)", "gquarter4": "quarter tiles up to a quarter round", "gstagger": "every second CU of an XCD starts late (grad)", "tickets3": "tickets from three rounds", "tickets2": "tickets from two rounds"}[knob]
if "ilv=on" in sys.argv:       # (other knobs measured with the interleaved div in place)
    _hip.set_div_interleave(1 << 40)
args = [a for a in sys.argv[1:] if not a.startswith("knob=") and a != "ilv=on"]
names = [a for a in args if a in ALL] or list(ALL)
sizes = [int(float(a)) for a in args if a not in ALL] or [20_000, 50_000, 98_304, 100_000, 131_072, 200_000, 400_000, 1_000_000]


def stages_of(what, E):
    g = torch.Generator(device="cuda").manual_seed(3)

    def inputs(expr):
        return {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64,
                              device="cuda", generator=g) for n in sorted(expr.all_args)}
    grad, div, fm = dg.grad(), dg.div(), dg.face_mass(4)
    gd = inputs(grad)
    dd = dict(inputs(div), J=gd["J"], R=gd["R"])
    if what == "grad":
        return [(grad, gd)]
    if what == "div":
        return [(div, dd)]
    if what == "graddiv":
        return [(div, dd), (grad, gd)]
    return [(div, dd), (grad, gd), (fm, inputs(fm))]


def timed(op, n):
    op.time_batch(10)
    return sorted(op.time_batch(n) / n for _ in range(5))[2]


for what in names:
    for E in sizes:
        stages = stages_of(what, E)
        nbytes = sum(measure._get_footprint_gbytes(e, E) * 1e9 for e, _ in stages) - (8.0 * (9 * E + 3 * 35 * 35) if len(stages) > 1 else 0.0)
        outs = [measure.generate_out_arrays(0, e, E, split=True) for e, _ in stages]
        op = f.bind_operator(stages, 0, out_dicts=outs)
        n = max(20, min(400, int(4e7 / E)))
        setter(0)
        timed(op, 5 * n)   # settle
        best = {0: 1e9, 1: 1e9}
        for rep in range(3):
            for mode in (0, 1):
                setter((1 << 40) if mode else 0)
                best[mode] = min(best[mode], timed(op, n))
        setter(0)
        op.launch(); torch.cuda.synchronize()
        ref = [{k: v.clone() for k, v in od.items()} for od in outs]
        for od in outs:
            for v in od.values():
                v.fill_(float("nan"))
        setter(1 << 40)
        op.launch(); torch.cuda.synchronize()
        info = _hip.last_launch_info()
        setter(0)
        same = all(torch.equal(od[k], rd[k]) for od, rd in zip(outs, ref) for k in od)
        a, b = best[0], best[1]
        print(f"{what:8s} E={E:8d}: default {a * 1e6:7.2f} us = {nbytes / a / 8e12:.3f}   {label} {b * 1e6:7.2f} us = {nbytes / b / 8e12:.3f}   "
              f"({(b / a - 1) * 100:+.1f} %)   same bits {same}   [{'dynamic' if info.get('dynamic_walk') else 'static'} walk, {info.get('kind')}]", flush=True)
        del op, outs, ref, stages
