"""p = 5 (Np = 56) MFMA kernels: one torch allocation per array against outputs from the split allocator (feinsum_amd.placement; round 2 compared against its arena scan, removed in round 4).

    python tools/bench_p5_placement.py [E]
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cases = [("grad", dg.grad(56)), ("div", dg.div(56)), ("face-mass x4", dg.face_mass(4, Np=56, Nfp=21)),
         ("p4 grad", dg.grad(35)), ("p4 face-mass x4", dg.face_mass(4))]
for name, expr in cases:
    gops = f.count_ops(expr, long_dim_length=E) * 1e-9
    roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", E)[np.dtype("float64")]
    row = []
    for what, tr in (("separate", {"variant": "mfma", "placement": "separate"}), ("split", {"variant": "mfma", "placement": "split"}),
                     ("separate", {"variant": "mfma", "placement": "separate"})):
        r = f.timeit_details(expr, cq=0, transform=tr, long_dim_length=E, min_secs=0.5)
        row.append(f"{what} {r.seconds_device * 1e3:7.4f} ms {gops / r.seconds_device:7.0f} GF/s ({gops / r.seconds_device / roof * 100:4.1f} %)")
    print(f"Np = 56 {name:14s} " if not name.startswith("p4") else f"Np = 35 {name[3:]:14s} ", " | ".join(row), flush=True)
