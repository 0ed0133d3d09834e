#!/usr/bin/env python
"""
What an ``evaluate()`` call costs on the host when it allocates its own outputs: per call, microseconds of host
time and of device time, for grad p = 4 with (a) outputs handed in, (b) outputs from the split allocator, recycled (default),
(c) the same with recycling off (round 4's behaviour: VMM map per call, unmap + device synchronisation on release), (d) outputs
from torch (``placement: separate``).

    python tools/evaluate_loop_cost.py [E=1000000] [calls=200]
"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd import measure, placement  # noqa: E402

E = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
expr = dg.grad()
host = measure.generate_host_input_arrays(expr, E, np_seed=0)
dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
fixed = measure.generate_out_arrays(0, expr, E, split=True)


def loop(label, **kw):
    for _ in range(5):
        out = f.evaluate(expr, 0, dev, **kw)
    torch.cuda.synchronize()
    va0 = placement.split_stats(0)["address_space_reserved"]
    t0 = time.perf_counter()
    for _ in range(calls):
        out = f.evaluate(expr, 0, dev, **kw)        # asynchronous; the previous output is dropped here
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    va = placement.split_stats(0)["address_space_reserved"] - va0
    print(f"{label:58s} host {t_host / calls * 1e6:8.1f} us per call   all {t_all / calls * 1e6:8.1f} us per call   address space +{va / 2**20:9.1f} MiB", flush=True)
    del out


print(f"# grad p=4, E={E}, {calls} evaluate() calls back to back (the kernel alone: ~{1192e-6 * E / 6.3:.0f} us)")
loop("outputs handed in (out_dict)", out_dict=fixed)
loop("own outputs: split allocator, recycled (default)")
os.environ["FEINSUM_SPLIT_RECYCLE_MIB"] = "0"
placement.recycle_trim(0)
loop("own outputs: split allocator, NOT recycled (round 4)")
del os.environ["FEINSUM_SPLIT_RECYCLE_MIB"]
loop("own outputs: torch (placement: separate)", transform={"placement": "separate"})
print(placement.recycle_stats())
