"""What the elements behind the last full tile cost: launch times at E and E + 1 ... 15 (timeit protocol, min of 3).
   python tools/ragged_cost.py"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root))
import torch  # noqa: F401
import dg
import feinsum_amd as f
from feinsum_amd import measure

import sys as _s
for name, expr in [c for c in (("grad", dg.grad()), ("div", dg.div()), ("face_mass x 4", dg.face_mass(4))) if not _s.argv[1:] or c[0].split()[0] in _s.argv[1:]]:
    for base in (20_000, 100_000, 1_000_000):
        row = []
        for extra in (0, 1, 3, 8, 15):
            t = min(measure.timeit_details(expr, cq=0, long_dim_length=base + extra, min_secs=0.2).seconds_device for _ in range(3))
            row.append(f"+{extra}: {t * 1e6:7.2f}")
        print(f"{name:14s} E = {base:7d}  " + "  ".join(row) + "  us", flush=True)
