"""Gradient einsum of the 3-D DG wave operator, p = 4, against its roofline.

Counterpart of the reference's ``examples/dg_wave_grad.py`` (``get_grad_einsum`` ``:12-24``,
``main`` ``:411-424``): the einsum is built the same way; where the reference passes a loopy
transformation (``paranumal_transform``), the hand-written gfx950 kernel is selected
(``transform=None`` = best available variant).

    python examples/dg_wave_grad.py [long_dim_length]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import feinsum_amd as f  # noqa: E402


def get_grad_einsum(ndofs, ndim):
    return f.einsum("xre,rij,ej->xei",
                    f.array("J", (ndim, ndim, "Nel")),
                    f.array("R", (ndim, ndofs, ndofs)),
                    f.array("u", ("Nel", ndofs)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    expr = get_grad_einsum(ndofs=35, ndim=3)
    print(expr)
    print(f.stringify_comparison_vs_roofline(expr, cq=0, transform=None, long_dim_length=n,
                                             ignore_unknown_device=True))


if __name__ == "__main__":
    main()
