#!/usr/bin/env python
"""
A/B in one process: the persistent walk with a static tail (rounds < 0) against dynamic tails of 1 ... 12 rounds
(fe_set_tail_rounds), on the SAME arrays -- outputs from the split allocator and from torch allocations.

    python tools/tail_ab.py [grad|div|facemass|grad5|div5|facemass5] [E=1000000] [rounds="-1 1 2 3 6 12"]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip, measure, placement  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "grad"
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
rounds = [int(r) for r in (sys.argv[3] if len(sys.argv) > 3 else "-1 1 2 3 6 12").split()]
def _tri(kind, Np=15, Nfp=5):
    J, R = f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np))
    if kind == "grad":
        return f.einsum("xre,rij,ej->xei", J, R, f.array("u", ("E", Np)))
    if kind == "div":
        return f.einsum("xre,rij,xej->ei", J, R, f.array("u", (2, "E", Np)))
    return f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)), f.array(f"v{k}", (3, "E", Nfp))]
                                               for k in range(3)])


expr = {"grad2": lambda: _tri("grad"), "div2": lambda: _tri("div"), "lift2": lambda: _tri("lift"),
        "grad2p2": lambda: _tri("grad", 6, 3), "div2p5": lambda: _tri("div", 21, 6),
        "bgrad3p3": lambda: dg.batched_grad(3, 20), "bdiv3p3": lambda: dg.batched_div(3, 20), "bgrad3p2": lambda: dg.batched_grad(3, 10),
        "fm5": lambda: dg.face_mass(5), "fm3p3": lambda: dg.face_mass(3, 20, 4, 10),
        "grad": dg.grad, "div": dg.div, "facemass": lambda: dg.face_mass(4), "grad5": lambda: dg.grad(56), "div5": lambda: dg.div(56),
        "facemass5": lambda: dg.face_mass(4, Np=56, Nfp=21), "bdiv3": lambda: dg.batched_div(3), "bgrad3": lambda: dg.batched_grad(3)}[what]()
nbytes = measure._get_footprint_gbytes(expr, E) * 1e9
flops = f.count_ops(expr, long_dim_length=E)
host = measure.generate_host_input_arrays(expr, E, np_seed=0)
dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}


def timed(bound, q, n):
    bound.time_batch(10, q.stream_ptr)
    return sorted(bound.time_batch(n, q.stream_ptr) / n for _ in range(5))[2]


n = max(20, min(400, int(4e7 / E)))
for label, split in (("split allocator", True), ("torch allocations", False)):
    outs = measure.generate_out_arrays(0, expr, E, split=split)
    q, bound, _ = measure._bind(expr, 0, dev, outs, None)
    _hip.set_tail_rounds(-1)
    timed(bound, q, 5 * n)   # settle
    for rep in range(3):
        cells = []
        for r in rounds:
            _hip.set_tail_rounds(r)
            t = timed(bound, q, n)
            cells.append(f"{r:4d}: {t * 1e6:7.2f} us ({nbytes / t / 8e12 * 100:4.1f} % of HBM, {flops / t * 1e-12:5.1f} TFLOP/s)")
        print(f"{what} E={E} {label}: " + "  ".join(cells), flush=True)
    _hip.set_tail_rounds(-1)
    f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
    ref = {k: v.clone() for k, v in outs.items()}
    _hip.set_tail_rounds(1 << 20)
    f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
    print("  dynamic tail gives the same bits:", all(torch.equal(outs[k], ref[k]) for k in ref), flush=True)
