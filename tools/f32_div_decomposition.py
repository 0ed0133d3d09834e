"""What bounds the float32 div kernel: launches with parts of the tile work removed (experiment build only).
   FEINSUM_HIP_LIB=build/libfeinsum_hip_exp.so FEINSUM_F32_DBG=<bits> python tools/f32_div_decomposition.py"""
import os, sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root))
import torch
import dg
import feinsum_amd as f
from feinsum_amd import _hip
from feinsum_amd.measure import generate_host_input_arrays

base = dg.div()
expr = f.batched_einsum(base.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in base.args])
E = 1_000_000
host = generate_host_input_arrays(expr, E, np_seed=1)
from feinsum_amd import placement
dev = {}
for k, v in host.items():     # every array from the split allocator, as timeit's
    dev[k] = placement.empty(v.shape, torch.float32, "cuda:0", written=False)
    dev[k].copy_(torch.from_numpy(v))
outs = {n: placement.empty((E, 35), torch.float32, "cuda:0") for n in expr.output_names}
dbg = int(os.environ.get("FEINSUM_F32_DBG", "0"))
for walk in ((-1, 1 << 20) if dbg == 0 else (-1,)):
    _hip.set_tail_rounds(walk)
    for _ in range(20):
        f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        for _ in range(50):
            f.evaluate(expr, 0, dev, out_dict=outs)
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / 50 * 1e3)
    print(f"dbg={os.environ.get('FEINSUM_F32_DBG', '0')} walk={'static' if walk < 0 else 'tickets'}: {best:.2f} us", flush=True)
