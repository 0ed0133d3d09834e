#!/usr/bin/env python
"""
A/B in one process: the streamed operand fetched with non-temporal loads (fe_set_temporal_loads_mib(0)) against plain loads
whenever the launch's inputs are at most N MiB, on the SAME arrays (outputs from the split allocator), back-to-back launches
as in the reference's timing protocol (src/feinsum/measure.py:248-275).

    python tools/temporal_ab.py [grad|div|facemass|graddiv|pipeline ...] [--sizes "2e4 5e4 1e5 2e5 5e5 1e6"] [--mib "0 248 1048576"]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="*", default=["grad", "div", "facemass"])
ap.add_argument("--sizes", default="2e4 5e4 1e5 2e5 5e5 1e6")
ap.add_argument("--mib", default="0 248 1048576")
args = ap.parse_args()
settings = [int(m) for m in args.mib.split()]


def stages_of(what, E):
    mk = lambda expr, seed: {k: torch.from_numpy(v).cuda() for k, v in measure.generate_host_input_arrays(expr, E, np_seed=seed).items()}  # noqa: E731
    if what in ("grad", "div", "facemass"):
        expr = {"grad": dg.grad, "div": dg.div, "facemass": lambda: dg.face_mass(4)}[what]()
        return [(expr, mk(expr, 0))]
    g, d = dg.grad(), dg.div()
    gd, dd = mk(g, 0), mk(d, 1)
    dd["J"], dd["R"] = gd["J"], gd["R"]
    st = [(d, dd), (g, gd)]
    if what == "pipeline":
        fm = dg.face_mass(4)
        st.append((fm, mk(fm, 2)))
    return st


def timed(op, n):
    op.time_batch(10)
    return sorted(op.time_batch(n) / n for _ in range(5))[2]


before = _hip.set_temporal_loads_mib(0)
for what in args.what:
    for E in [int(float(s)) for s in args.sizes.split()]:
        stages = stages_of(what, E)
        nbytes = sum(measure._get_footprint_gbytes(e, E) * 1e9 for e, _ in stages)
        if what != "grad" and len(stages) > 1:      # J and D are shared by div and grad
            nbytes -= 8.0 * (9 * E + 3 * 35 * 35)
        in_mib = sum(t.numel() * 8 for _, d in stages for t in {id(t): t for t in d.values()}.values()) / 2**20
        outs = [measure.generate_out_arrays(0, e, E, split=True) for e, _ in stages]
        op = f.bind_operator(stages, 0, out_dicts=outs)
        n = max(20, min(400, int(4e7 / E)))
        timed(op, 3 * n)
        for rep in range(3):
            cells = []
            for mib in settings:
                _hip.set_temporal_loads_mib(mib)
                t = timed(op, n)
                cells.append(f"<= {mib} MiB: {t * 1e6:8.2f} us ({nbytes / t / 8e12 * 100:4.1f} %)")
            print(f"{what} E={E} (inputs {in_mib:.0f} MiB, {len(op.launches)} launch): " + "   ".join(cells), flush=True)
        _hip.set_temporal_loads_mib(0)
        op.launch(); op.queue.finish()
        ref = [{k: v.clone() for k, v in od.items()} for od in outs]
        _hip.set_temporal_loads_mib(1 << 20)
        op.launch(); op.queue.finish()
        print("  same bits:", all(torch.equal(od[k], rd[k]) for od, rd in zip(outs, ref) for k in rd), flush=True)
        del op, outs, stages, ref
_hip.set_temporal_loads_mib(before)
