"""MFMA kernels of all tetrahedral orders: every array from torch ("separate") against outputs from the split allocator
("split", timeit's default; round 2's arena scan, "tuned", was removed in round 4), and the split placement once more
with the static walk (fe_set_tail_rounds(-1)).

    python tools/bench_orders_placement.py [E]
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import _hip  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for Np, Nfp in ((4, 3), (10, 6), (20, 10), (35, 15), (56, 21)):
    for name, expr in (("grad", dg.grad(Np)), ("div", dg.div(Np)), ("face-mass x4", dg.face_mass(4, Np=Np, Nfp=Nfp))):
        gops = f.count_ops(expr, long_dim_length=E) * 1e-9
        roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", E)[np.dtype("float64")]
        row = []
        modes = ["separate", "split"]
        for what, tr in ((m, {"variant": "mfma", "placement": m}) for m in modes):
            r = f.timeit_details(expr, cq=0, transform=tr, long_dim_length=E, min_secs=0.5)
            row.append(f"{what} {r.seconds_device * 1e3:7.4f} ms {gops / r.seconds_device:7.0f} GF/s ({gops / r.seconds_device / roof * 100:4.1f} %)")
        _hip.set_tail_rounds(-1)
        r = f.timeit_details(expr, cq=0, transform={"variant": "mfma", "placement": "split"}, long_dim_length=E, min_secs=0.5)
        _hip.set_tail_rounds(1 << 20)
        row.append(f"split, static walk {r.seconds_device * 1e3:7.4f} ms ({gops / r.seconds_device / roof * 100:4.1f} %)")
        print(f"Np = {Np:2d} {name:14s} " + " | ".join(row), flush=True)
