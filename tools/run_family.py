#!/usr/bin/env python
"""Launch one einsum family a number of times (the program to put under rocprofv3):
    python3 tools/run_family.py <grad|div|facemass|divcomp|crossprod> <Np> <E> <launches> [variant]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import measure  # noqa: E402

fam, Np, E, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
variant = sys.argv[5] if len(sys.argv) > 5 else None
Nfp = {4: 3, 10: 6, 20: 10, 35: 15, 56: 21}[Np]
expr = {"grad": lambda: dg.grad(Np), "div": lambda: dg.div(Np), "facemass": lambda: dg.face_mass(4, Np=Np, Nfp=Nfp),
        "divcomp": lambda: dg.batched_div_components(Np), "crossprod": lambda: dg.cross_product_batch(Np)}[fam]()
q = f.DeviceQueue(0)
g = torch.Generator(device="cuda").manual_seed(0)
dev = {name: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[name]),
                        dtype=torch.float64, device="cuda", generator=g) for name in sorted(expr.all_args)}
_, bound, _ = measure._bind(expr, q, dev, None, variant)
for _ in range(5):
    bound.launch(q.stream_ptr)
q.finish()
t = bound.time_batch(n, q.stream_ptr) / n
flops = f.count_ops(expr, long_dim_length=E)
print(f"{fam} Np={Np} E={E}: {t * 1e3:.4f} ms  {flops / t * 1e-9:.0f} GFLOP/s")
