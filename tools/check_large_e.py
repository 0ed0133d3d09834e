import sys, time, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, feinsum_amd as f, dg
from oracle import np_oracle, c_oracle
E = 8_000_000
g = torch.Generator(device="cuda").manual_seed(3)
J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
D = torch.rand((3, 35, 35), dtype=torch.float64, device="cuda", generator=g)
u = torch.rand((E, 35), dtype=torch.float64, device="cuda", generator=g)
out = f.evaluate(dg.grad(), 0, {"J": J, "R": D, "u": u}, wait=True)["_fe_out"]
worst = 0.0
for s in (0, 4_999_984, E - 1000):
    sl = slice(s, s + 1000)
    ref = np_oracle.reference_outputs("xre,rij,ej->xei", [[J[:, :, sl].cpu().numpy(), D.cpu().numpy(), u[sl].cpu().numpy()]])[0]
    worst = max(worst, np_oracle.max_rel_err(out[:, sl].cpu().numpy(), ref))
r = f.timeit_details(dg.grad(), cq=0, long_dim_length=E, min_secs=0.3, validate=False)
print(f"grad E=8e6: max rel err {worst:.2e}; {r.seconds_device*1e3:.3f} ms -> {7980*E/r.seconds_device*1e-9:.0f} GFLOP/s")
print("affinity cpus:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
