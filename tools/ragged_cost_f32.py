import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import dg
import feinsum_amd as f
from feinsum_amd import measure
def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])
for name, expr in (("grad", f32(dg.grad())), ("div", f32(dg.div())), ("face_mass", f32(dg.face_mass(4)))):
    for base in (20_000, 100_000):
        row = []
        for extra in (0, 4, 12):
            t = min(measure.timeit_details(expr, cq=0, long_dim_length=base + extra, min_secs=0.2).seconds_device for _ in range(3))
            row.append(f"+{extra}: {t*1e6:7.2f}")
        print(f"float32 {name:10s} E = {base:7d}  " + "  ".join(row) + "  us", flush=True)
