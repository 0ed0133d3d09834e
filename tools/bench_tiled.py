"""The LDS-tiled VALU kernel against the plain generic kernels (and MFMA where compiled).

    python tools/bench_tiled.py [E]
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def tri(Np, Nfp):
    grad2 = f.einsum("xre,rij,ej->xei", f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array("u", ("E", Np)))
    div2 = f.einsum("xre,rij,xej->ei", f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array("u", (2, "E", Np)))
    lift2 = f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)),
                                                 f.array(f"v{k}", (3, "E", Nfp))] for k in range(3)])
    return grad2, div2, lift2


cases = []
for Np, Nfp in ((56, 21), (35, 15)):
    cases += [(f"3D Np={Np} grad", dg.grad(Np)), (f"3D Np={Np} div", dg.div(Np)),
              (f"3D Np={Np} face-mass x4", dg.face_mass(4, Np=Np, Nfp=Nfp))]
for Np, Nfp in ((15, 5), (6, 3)):
    g2, d2, l2 = tri(Np, Nfp)
    cases += [(f"2D Np={Np} grad", g2), (f"2D Np={Np} div", d2), (f"2D Np={Np} lift x3", l2)]
for name, expr in cases:
    row = []
    for v in ("tiled", "generic", "mfma"):
        try:
            r = f.timeit_details(expr, cq=0, transform=v, long_dim_length=E, min_secs=0.2)
        except NotImplementedError:
            row.append(f"{v} n/a")
            continue
        gops = f.count_ops(expr, long_dim_length=E) * 1e-9
        roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", E)[np.dtype("float64")]
        row.append(f"{v} {r.seconds_device * 1e3:7.3f} ms {gops / r.seconds_device:7.0f} GF/s ({gops / r.seconds_device / roof * 100:4.1f} %)")
    print(f"{name:24s} " + " | ".join(row), flush=True)
