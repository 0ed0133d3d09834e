import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dg, numpy as np
import feinsum_amd as f
from feinsum_amd import measure
def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])
e32 = f32(dg.grad())
for E in (200_000, 300_000, 400_000, 500_000, 700_000):
    t = measure.timeit_details(e32, cq=0, long_dim_length=E, min_secs=0.3)
    print(f"grad float32 E={E}: {t.seconds_device*1e6:.2f} us", flush=True)
