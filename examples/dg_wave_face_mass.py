"""Face-mass (lift) einsum of the 3-D DG wave operator, p = 4, four face fields
(reference: ``examples/dg_wave_face_mass.py``; ``ifj,fe,fej->ei`` as in
``tuning/impls/ifj_fe_fej_to_ei.py:46-60``).

    python examples/dg_wave_face_mass.py [long_dim_length]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import feinsum_amd as f  # noqa: E402


def get_face_mass_einsum(nfaces, nvoldofs, nfacedofs, nfields):
    return f.batched_einsum(
        "ifj,fe,fej->ei",
        [[f.array("L", (nvoldofs, nfaces, nfacedofs)), f.array("J", (nfaces, "Nel")),
          f.array(f"v{k}", (nfaces, "Nel", nfacedofs))] for k in range(nfields)])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    expr = get_face_mass_einsum(nfaces=4, nvoldofs=35, nfacedofs=15, nfields=4)
    print(f.stringify_comparison_vs_roofline(expr, cq=0, transform=None, long_dim_length=n))


if __name__ == "__main__":
    main()
