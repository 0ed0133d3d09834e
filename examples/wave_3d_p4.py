"""The whole 3-D wave operator -- div(v), grad(u) and the lift of four face fields -- from one
description, as ``examples/wave_3d_p4_auto.py:16-63`` of the reference: there the three parts sit
in one loopy kernel separated by global barriers and get their transforms from the archive
(``f.query`` / ``:119-139``); here the three einsums are bound together and run as ONE launch.

    python examples/wave_3d_p4.py [n_elements]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import feinsum_amd as f  # noqa: E402

NDOFS, NFACEDOFS, NFACES = 35, 15, 4


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    div = f.einsum("xre,rij,xej->ei", f.array("J", (3, 3, "Nel")), f.array("D", (3, NDOFS, NDOFS)),
                   f.array("v", (3, "Nel", NDOFS)))
    grad = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "Nel")), f.array("D", (3, NDOFS, NDOFS)),
                    f.array("u", ("Nel", NDOFS)))
    lift = f.batched_einsum("ifj,fe,fej->ei", [[f.array("L", (NDOFS, NFACES, NFACEDOFS)), f.array("Jface", (NFACES, "Nel")),
                                                f.array(f"F_{k}", (NFACES, "Nel", NFACEDOFS))] for k in range(4)])
    q = f.DeviceQueue(0)
    arrays = [dict(f.generate_input_arrays(q, e, n, np_seed=k)) for k, e in enumerate((div, grad, lift))]
    arrays[1]["J"], arrays[1]["D"] = arrays[0]["J"], arrays[0]["D"]     # one geometry, one operator
    stages = list(zip((div, grad, lift), arrays))

    op = f.bind_operator(stages, q)
    print("launches per evaluation:", op.entry_points)
    op.launch()
    q.finish()
    # validate one element batch of every stage against numpy, as validate_batched_einsum_transform does
    for expr, arr, outs in zip((div, grad, lift), arrays, op.outputs):
        for name, row in zip(expr.output_names, expr.args):
            sl = {a.name: arr[a.name].cpu().numpy() for a in row}
            ref = np.einsum(expr.get_subscripts(), *[sl[a.name] for a in row], optimize="optimal")
            np.testing.assert_allclose(outs[name].cpu().numpy(), ref, rtol=1e-10, atol=1e-10)
    for _ in range(5):
        op.launch()
    secs = op.time_batch(20) / 20
    gops = sum(f.count_ops(e, long_dim_length=n) for e in (div, grad, lift)) * 1e-9
    print(f"{n} elements: {secs * 1e3:.4f} ms per operator evaluation, {gops / secs:.0f} GFLOP/s")
    del torch


if __name__ == "__main__":
    main()
