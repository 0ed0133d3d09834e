"""A few launches of one float32 einsum for a --pmc pass (tools/f32_pmc.sh):  python3 tools/f32_pmc_run.py div 1000000"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root))
import torch
import dg
import feinsum_amd as f
from feinsum_amd.measure import generate_host_input_arrays

name, E = sys.argv[1], int(sys.argv[2])
base = {"grad": dg.grad, "div": dg.div, "face_mass": lambda: dg.face_mass(4)}[name]()
expr = f.batched_einsum(base.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in base.args])
host = generate_host_input_arrays(expr, E, np_seed=1)
dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
for _ in range(12):
    f.evaluate(expr, 0, dev, wait=True)
