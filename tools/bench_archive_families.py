"""The remaining einsum families of the reference's archive (data/transform_archive_v5.sqlite keys
'a,ab,cb->ac', 'ba,ca->bc', 'ab,b->a', 'ab->a', 'ab->ab', 'cab->cab'), as spelled in
tuning/impls/{e_ij_ej_to_ei_no_prftch,ij_ej_to_ei_no_prftch,ij_j_to_i,ij_to_i,ij_ij_to_ij,ijk_ijk_to_ijk}.py.

    python tools/bench_archive_families.py [E]
"""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import feinsum_amd as f  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Np = 35
cases = {
    "e,ij,ej->ei x4": f.batched_einsum("e,ij,ej->ei", [[f.array("J", ("E",)), f.array("D", (Np, Np)),
                                                        f.array(f"u{k}", ("E", Np))] for k in range(4)]),
    "ij,ej->ei": f.einsum("ij,ej->ei", f.array("D", (Np, Np)), f.array("u", ("E", Np))),
    "ej,j->e": f.einsum("ej,j->e", f.array("A", ("E", Np)), f.array("w", (Np,))),
    "ej->e": f.einsum("ej->e", f.array("A", ("E", Np))),
    "ej,ej->ej": f.einsum("ej,ej->ej", f.array("A", ("E", Np)), f.array("B", ("E", Np))),
    "fej,fej->fej": f.einsum("fej,fej->fej", f.array("A", (4, "E", 15)), f.array("B", (4, "E", 15))),
}
for n in (56, 20, 10, 4):
    cases[f"e,ij,ej->ei x4 Np={n}"] = f.batched_einsum("e,ij,ej->ei", [[f.array("J", ("E",)), f.array("D", (n, n)),
                                                                       f.array(f"u{k}", ("E", n))] for k in range(4)])
for name, expr in cases.items():
    r = f.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.3)
    gops = f.count_ops(expr, long_dim_length=E) * 1e-9
    roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", E)[np.dtype("float64")]
    print(f"{name:24s} {r.seconds_device * 1e3:8.4f} ms  {gops / r.seconds_device:9.0f} GFLOP/s  roofline {roof:8.0f}"
          f" -> {gops / r.seconds_device / roof * 100:5.1f} %")
