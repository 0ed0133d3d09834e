"""Divergence of the 3-D DG wave operator, p = 4: the three-component batch the reference's
``examples/dg_wave_div.py:13-25`` builds (``es,sij,ej->ei`` for x, y, z) and the fused form
``xre,rij,xej->ei`` its tuned transforms target (``tuning/impls/xre_rij_xej_to_ei.py``).

    python examples/dg_wave_div.py [long_dim_length]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import feinsum_amd as f  # noqa: E402


def get_div_einsum(ndofs, ndim):
    return f.batched_einsum(
        "es, sij, ej -> ei",
        [[f.array("J" + c, ("Nel", ndim)), f.array("R", (ndim, ndofs, ndofs)), f.array("u" + c, ("Nel", ndofs))]
         for c in "xyz"[:ndim]])


def get_fused_div_einsum(ndofs, ndim):
    return f.einsum("xre,rij,xej->ei", f.array("J", (ndim, ndim, "Nel")), f.array("R", (ndim, ndofs, ndofs)),
                    f.array("v", (ndim, "Nel", ndofs)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for expr in (get_div_einsum(35, 3), get_fused_div_einsum(35, 3)):
        print(expr.get_subscripts(), f"x {expr.b}")
        print(f.stringify_comparison_vs_roofline(expr, cq=0, transform=None, long_dim_length=n))


if __name__ == "__main__":
    main()
