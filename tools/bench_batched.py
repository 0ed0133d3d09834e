"""Batched grad / div (b fields sharing J and D in one launch) against b single-field launches.

    python tools/bench_batched.py [E]
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
single = {}
for name, expr in (("grad", dg.grad()), ("div", dg.div())):
    single[name] = f.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.3).seconds_device
    print(f"{name} x1: {single[name] * 1e3:.4f} ms")
for b in (2, 3, 5, 6, 8):
    for name, expr in (("grad", dg.batched_grad(b)), ("div", dg.batched_div(b))):
        r = f.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.3)
        gops = f.count_ops(expr, long_dim_length=E) * 1e-9
        roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", E)[np.dtype("float64")]
        print(f"batched {name} b={b}: {r.seconds_device * 1e3:.4f} ms = {r.seconds_device / (b * single[name]):.3f} x"
              f" (b single launches)  {gops / r.seconds_device:.0f} GFLOP/s, roofline {roof:.0f}"
              f" -> {gops / r.seconds_device / roof * 100:.1f} %")
