"""float32 grad / div of tetrahedra p = 1 ... 4 through timeit: the MFMA kernels (templated on Np: grad round 4, div round 5) against the
tiled VALU kernel in float.
    python tools/bench_f32_orders.py [E] [grad div facemass]
"""
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dg, numpy as np
import feinsum_amd as f
from feinsum_amd import measure, _hip
def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])
E = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
fams = [a for a in sys.argv[2:] if a in ("grad", "div", "facemass")] or ["grad"]
NFP = {4: 3, 10: 6, 20: 10, 35: 15}
for fam, Np in [(fam, Np) for fam in fams for Np in (4, 10, 20, 35)]:
    e32 = f32(dg.grad(Np) if fam == "grad" else dg.div(Np) if fam == "div" else dg.face_mass(4, Np=Np, Nfp=NFP[Np]))
    cells = []
    for variant in ("auto", "tiled"):
        t = measure.timeit_details(e32, cq=0, long_dim_length=E, min_secs=0.4, transform=variant)
        gops = f.count_ops(e32, long_dim_length=E) * 1e-9
        roof = f.get_roofline_flop_rate(e32, "AMD Instinct MI355X", E)[np.dtype("float32")]
        cells.append(f"{variant}: {t.seconds_device*1e3:.4f} ms {gops / t.seconds_device:.0f} GFLOP/s = {gops / t.seconds_device / roof * 100:.1f} % of {roof:.0f}")
    print(f"{fam} float32 Np={Np} E={E}: " + " | ".join(cells), flush=True)
for l in _hip.kernel_resources().splitlines():
    if "float32" in l:
        print(l)
