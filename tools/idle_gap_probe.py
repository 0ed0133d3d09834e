#!/usr/bin/env python
"""
What an idle gap in front of a short timed region costs: the grad launch at E = 1e6 (allocator outputs), 20 launches timed by HIP
events, after an idle gap of 0 ... 20 ms on the host (the GPU has nothing queued during the gap).

    python tools/idle_gap_probe.py
"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import measure  # noqa: E402

E = 1_000_000
expr = dg.grad()
g = torch.Generator(device="cuda").manual_seed(3)
dev = {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64, device="cuda", generator=g)
       for n in sorted(expr.all_args)}
outs = measure.generate_out_arrays(0, expr, E, split=True)
op = f.bind_operator([(expr, dev)], 0, out_dicts=[outs])
op.time_batch(600)
for gap_ms in (0.0, 0.05, 0.2, 0.5, 1.0, 2.0, 5.0, 20.0, 0.0):
    res = []
    for rep in range(5):
        op.time_batch(100)                      # busy
        torch.cuda.synchronize()
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        res.append(op.time_batch(20) / 20)
    res.sort()
    print(f"idle gap {gap_ms:5.2f} ms: 20 launches at {res[2] * 1e6:7.2f} us each (min {res[0] * 1e6:7.2f}, max {res[-1] * 1e6:7.2f})", flush=True)
