import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dg, numpy as np
import feinsum_amd as f
from feinsum_amd import measure, _hip
def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])
e32 = f32(dg.grad())
for E in (100_000, 1_000_000, 4_000_000):
    for pl in ("split", "separate"):
        t = measure.timeit_details(e32, cq=0, long_dim_length=E, min_secs=0.5, transform={"placement": pl})
        gops = f.count_ops(e32, long_dim_length=E) * 1e-9
        roof = f.get_roofline_flop_rate(e32, "AMD Instinct MI355X", E)[np.dtype("float32")]
        print(f"grad float32 E={E} {pl}: {t.seconds_device*1e3:.4f} ms {gops / t.seconds_device:.0f} GFLOP/s {gops / t.seconds_device / roof * 100:.1f} % of {roof:.0f}", flush=True)
print([l for l in _hip.kernel_resources().splitlines() if "float32" in l])
