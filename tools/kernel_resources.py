#!/usr/bin/env python
"""
Registers, LDS and resident blocks per CU of every MFMA / tiled kernel of the library, as the HIP
runtime reports them on this device (run on the GPU box):

    python tools/kernel_resources.py > profiles/r02/resources.txt

Each kernel family is launched once on a small batch (a kernel is configured, and its residency
checked against what its persistent grid assumes, on its first launch: configure_kernel in
feinsum_hip.hip); the table is then read back through fe_kernel_resources().
"""

from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main() -> None:
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import _hip
    from feinsum_amd.measure import generate_host_input_arrays

    E = 2048
    exprs = []
    for Np, Nfp in ((4, 3), (10, 6), (20, 10), (35, 15), (56, 21)):
        exprs += [dg.grad(Np), dg.div(Np), dg.batched_div_components(Np), dg.face_mass(4, Np=Np, Nfp=Nfp),
                  dg.face_mass(2, Np=Np, Nfp=Nfp), dg.mass_apply(2, Np), dg.operator_apply(Np)]
        exprs.append(dg.cross_product_batch(Np))
    exprs += [dg.grad(84)]                                    # tiled kernel
    for expr in exprs:
        host = generate_host_input_arrays(expr, E)
        f.evaluate(expr, 0, {k: torch.from_numpy(v).cuda() for k, v in host.items()}, wait=True)
    # the fused launches
    for Np, Nfp in ((4, 3), (10, 6), (20, 10), (35, 15)):
        stages = []
        for k, expr in enumerate((dg.div(Np), dg.grad(Np), dg.face_mass(4, Np=Np, Nfp=Nfp))):
            host = generate_host_input_arrays(expr, E, np_seed=k)
            stages.append((expr, {n: torch.from_numpy(v).cuda() for n, v in host.items()}))
        stages[1][1]["J"], stages[1][1]["R"] = stages[0][1]["J"], stages[0][1]["R"]
        f.evaluate_operator(stages[:2], 0, wait=True)
        f.evaluate_operator(stages, 0, wait=True)
    # triangles
    import feinsum_amd as f2
    for Np, Nfp in ((3, 2), (6, 3), (10, 4), (15, 5), (21, 6)):
        g = f2.einsum("xre,rij,ej->xei", f2.array("J", (2, 2, "E")), f2.array("R", (2, Np, Np)), f2.array("u", ("E", Np)))
        d = f2.einsum("xre,rij,xej->ei", f2.array("J", (2, 2, "E")), f2.array("R", (2, Np, Np)), f2.array("u", (2, "E", Np)))
        lift = f2.batched_einsum("ef,fij,fej->ei", [[f2.array("J", ("E", 3)), f2.array("R", (3, Np, Nfp)),
                                                     f2.array(f"v{k}", (3, "E", Nfp))] for k in range(3)])
        for expr in (g, d, lift):
            host = generate_host_input_arrays(expr, E)
            f.evaluate(expr, 0, {k: torch.from_numpy(v).cuda() for k, v in host.items()}, wait=True)
    name = f.DeviceQueue(0).device.name
    print(f"# {name}; kernels as compiled into feinsum_amd/libfeinsum_hip.so (hipFuncGetAttributes, "
          f"hipOccupancyMaxActiveBlocksPerMultiprocessor)")
    print(_hip.kernel_resources(), end="")


if __name__ == "__main__":
    main()
