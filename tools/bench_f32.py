"""float32 DG einsums at p = 4 (grad / div / face-mass x 4) through timeit: outputs from the split allocator and from torch
allocations, against the float32 min-roofline; then the registers of the float32 kernels.
    python tools/bench_f32.py [families...]
"""
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dg, numpy as np
import feinsum_amd as f
from feinsum_amd import measure, _hip
def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])
CASES = {"grad": dg.grad, "div": dg.div, "facemass": lambda: dg.face_mass(4)}
for name in (sys.argv[1:] or list(CASES)):
    e32 = f32(CASES[name]())
    for E in (100_000, 1_000_000, 4_000_000):
        for pl in ("split", "separate"):
            t = measure.timeit_details(e32, cq=0, long_dim_length=E, min_secs=0.5, transform={"placement": pl})
            gops = f.count_ops(e32, long_dim_length=E) * 1e-9
            roof = f.get_roofline_flop_rate(e32, "AMD Instinct MI355X", E)[np.dtype("float32")]
            print(f"{name} float32 E={E} {pl}: {t.seconds_device*1e3:.4f} ms {gops / t.seconds_device:.0f} GFLOP/s "
                  f"{gops / t.seconds_device / roof * 100:.1f} % of {roof:.0f}", flush=True)
for l in _hip.kernel_resources().splitlines():
    if "float32" in l:
        print(l)
