"""Measure the kernel variants on this device and write the timing-fact archive
(`feinsum_amd.sql_utils`, the reference's FEINSUM_TIMING_FACTS format).

    python tools/record_archive.py [out.sqlite] [long_dim_length]

The shipped `feinsum_amd/data/transform_archive_mi355x.sqlite` is the output of this script
on one MI355X at the reference's default long_dim_length = 100000
(src/feinsum/sql_utils.py:418), i.e. the size its own archive was recorded at.
"""
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")

import numpy as np  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/transform_archive_mi355x.sqlite"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
if os.path.exists(out):
    os.remove(out)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)

ORDERS = {4: 3, 10: 6, 20: 10, 35: 15}
cases = []
for Np, Nfp in ORDERS.items():
    cases += [(dg.grad(Np), ("mfma",)), (dg.div(Np), ("mfma",)), (dg.grad_t(Np), ("mfma",)), (dg.div_t(Np), ("mfma",)),
              (dg.face_mass(4, Np=Np, Nfp=Nfp), ("mfma",)), (dg.face_mass_ifj_fe(4, Np=Np, Nfp=Nfp), ("mfma",)),
              (dg.batched_div_components(Np), ("mfma",))]
cases += [(dg.grad(), ("generic",)), (dg.div(), ("generic",)), (dg.face_mass(), ("generic",)),
          (dg.face_mass(19), ("mfma",)), (dg.face_mass_jfi_fe(4), ("mfma",)),
          (dg.cross_product_batch(), ("mfma", "generic"))]
cases += [(dg.batched_grad(b), ("mfma",)) for b in (3, 5)] + [(dg.batched_div(b), ("mfma",)) for b in (3, 5, 6)]
# the other families of the reference's archive (generic einsum kernel unless noted)
cases += [(dg.mass_apply(b, Np), ("mfma",)) for Np in (4, 10, 20, 35) for b in (4, 16)]
cases += [(dg.operator_apply(Np), ("mfma",)) for Np in (3, 4, 6, 10, 15, 20, 35)]
cases += [(dg.mass_apply(4), ("generic",)), (dg.operator_apply(), ("generic",))]
for K in (4, 10, 20, 35):
    cases += [(f.einsum("ej,j->e", f.array("A", ("E", K)), f.array("w", (K,))), ("generic",)),
              (f.einsum("ej->e", f.array("A", ("E", K))), ("generic",)),
              (f.einsum("ej,ej->ej", f.array("A", ("E", K)), f.array("B", ("E", K))), ("generic",))]
cases += [(f.einsum("fej,fej->fej", f.array("A", (4, "E", K)), f.array("B", (4, "E", K))), ("generic",))
          for K in (3, 6, 10, 15)]

# triangles p = 1..5 and tetrahedra p = 5 (MFMA; the tiled VALU kernel beside it)
for Np, Nfp in ((3, 2), (6, 3), (10, 4), (15, 5), (21, 6)):
    cases += [(f.einsum("xre,rij,ej->xei", f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array("u", ("E", Np))), ("mfma",)),
              (f.einsum("xre,rij,xej->ei", f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array("u", (2, "E", Np))), ("mfma",)),
              (f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)),
                                                    f.array(f"v{k}", (3, "E", Nfp))] for k in range(3)]), ("mfma",))]
cases += [(dg.grad(56), ("mfma", "tiled", "generic")), (dg.div(56), ("mfma", "tiled")),
          (dg.face_mass(4, Np=56, Nfp=21), ("mfma", "tiled")),
          (dg.mass_apply(4, 56), ("mfma",)), (dg.grad(), ("tiled",)), (dg.div(), ("tiled",)), (dg.face_mass(), ("tiled",))]

q = f.DeviceQueue(0)
for expr, variants in cases:
    for v in variants:
        f.record_facts(expr, q, v, database=out, long_dim_length=E)
        best = max(f.query(expr, q.device, database=out), key=lambda k: k.giga_op_rate(np.float64))
        print(f"{expr.get_subscripts():22s} b={expr.b:2d} {dict(f.canonicalize_einsum(expr).index_to_dim_length)!s:60.60s}"
              f" {v:8s} best {best.giga_op_rate(np.float64):9.0f} GFLOP/s", flush=True)
print(f"{len(f.get_timed_einsums_in_db(q.device, database=out))} einsums in {out}")
