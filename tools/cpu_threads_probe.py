import sys, time, os
nt = sys.argv[1]
if nt != "default": os.environ["OMP_NUM_THREADS"] = nt
sys.path.insert(0, '.')
import numpy as np
from oracle import c_oracle
lib = c_oracle.load(native=True)
E = 200_000; rng = np.random.default_rng(0)
J, D, u = rng.random((3, 3, E)), rng.random((3, 35, 35)), rng.random((E, 35)); out = np.empty((3, E, 35))
lib.oracle_grad3d_hoisted(J, D, u, out, E, 35)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3: lib.oracle_grad3d_hoisted(J, D, u, out, E, 35); n += 1
dt = time.perf_counter() - t0
print(f"OMP threads {nt} ({lib.oracle_num_threads()}): {7980*E*n/dt*1e-9:.1f} GFLOP/s")
