"""grad, short launches: write-through against non-temporal output stores (fe_set_write_through_mib), timeit protocol.
   python tools/write_through_ab.py"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root))
import torch  # noqa: F401
import dg
import feinsum_amd as f
from feinsum_amd import _hip, measure

import sys as _sys
CASES = {"grad": dg.grad, "grad_p3": lambda: dg.grad(20), "div": dg.div, "face_mass": lambda: dg.face_mass(4), "div_p3": lambda: dg.div(20)}
for name in (_sys.argv[1:] or ["grad", "grad_p3"]):
    expr = CASES[name]()
    Np = name
    for E in (20_000, 50_000, 80_000, 100_000, 120_000, 140_000, 160_000):
        row = []
        for rep in range(2):
            for mib in (0, 1 << 20):
                _hip.set_write_through_mib(mib)
                row.append(min(measure.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.2).seconds_device for _ in range(3)) * 1e6)
        nt, wt = min(row[0], row[2]), min(row[1], row[3])
        print(f"{name} E={E:7d}: nt {nt:6.2f} us  write-through {wt:6.2f} us  ({(wt / nt - 1) * 100:+.1f} %)", flush=True)
