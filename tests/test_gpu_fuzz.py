"""Seeded random sweep (tools/fuzz_gpu.py) over families, shapes, layouts, field counts, element
counts (0, 1, around the tile sizes, ragged) and kernel variants against the oracle."""

import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_random_cases_all_variants():
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import fuzz_gpu

    assert fuzz_gpu.run(120, seed=7) == 0
    assert fuzz_gpu.run_operator(30, seed=7) == 0
    assert fuzz_gpu.run_einsum(120, seed=7) == 0


def test_random_dynamic_walk_cases():
    """Random large launches (all orders, batched, triangles, fused operators) on random streams, grid sizes and load modes:
    bitwise the static walk with non-temporal loads on the full grid; every ticket counter zero afterwards."""
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import fuzz_gpu

    assert fuzz_gpu.run_dynamic_walk(24, seed=11) == 0


def test_empty_batch_of_planes():
    # E = 0: empty tensors have no addresses; the planes launch must not mistake them for unwanted planes
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd.measure import generate_host_input_arrays

    expr = dg.cross_product_batch()
    host = generate_host_input_arrays(expr, 0)
    outs = f.evaluate(expr, 0, {k: torch.from_numpy(v).cuda() for k, v in host.items()}, wait=True)
    assert all(tuple(o.shape) == (0, 35) for o in outs.values())


@pytest.mark.timeout(120)
def test_generic_einsum_with_an_empty_summation_axis():
    """A summed index of extent 0 (the long axis summed away, E = 0): every output entry is an empty
    sum -- zeros, and the launch returns (the kernel's odometer must not spin on a zero extent)."""
    import torch

    import feinsum_amd as f

    for subs, shapes in (("ij,ej->i", [(5, 7), ("E", 7)]), ("ej,ej->j", [("E", 6), ("E", 6)]),
                         ("eij,ej->i", [("E", 3, 4), ("E", 4)])):
        args = [f.array(f"a{k}", s) for k, s in enumerate(shapes)]
        expr = f.einsum(subs, *args)
        dev = {a.name: torch.zeros(tuple(0 if d == "E" else d for d in s), dtype=torch.float64, device="cuda")
               for a, s in zip(args, shapes)}
        out = f.evaluate(expr, 0, dev, wait=True)["_fe_out"]
        assert out.numel() > 0 and bool((out == 0).all())


@pytest.mark.parametrize("variant", ["auto", "tiled", "generic"])
def test_kernels_write_only_their_outputs(variant):
    """Outputs placed between sentinel guard bands: nothing outside the output arrays is written
    (ragged element counts, every family, batched and fused launches)."""
    import numpy as np
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd.measure import generate_host_input_arrays

    guard = 4096
    exprs = [dg.grad(), dg.div(), dg.face_mass(4), dg.face_mass_ifj_fe(3), dg.batched_grad(3), dg.batched_div(2),
             dg.batched_div_components(), dg.cross_product_batch(), dg.mass_apply(4), dg.operator_apply(),
             dg.grad(20), dg.div(10), dg.face_mass(4, Np=4, Nfp=3), dg.grad(56), dg.face_mass(4, Np=56, Nfp=21)]
    for E in (17, 1003, 4111):
        for expr in exprs:
            host = generate_host_input_arrays(expr, E, np_seed=E)
            dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
            shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
            n = int(np.prod(shape))
            bufs, outs = [], {}
            for name in expr.output_names:
                buf = torch.full((n + 2 * guard,), -7.25, dtype=torch.float64, device="cuda")
                bufs.append(buf)
                outs[name] = buf[guard:guard + n].view(shape)
            try:
                f.evaluate(expr, 0, dev, out_dict=outs, transform=variant, wait=True)
            except NotImplementedError:      # e.g. "tiled" for the planes launch
                continue
            for buf in bufs:
                assert bool((buf[:guard] == -7.25).all()) and bool((buf[guard + n:] == -7.25).all()), (expr.get_subscripts(), E)
                assert bool(torch.isfinite(buf[guard:guard + n]).all()) and not bool((buf[guard:guard + n] == -7.25).any())
    # the fused operator launches too
    E = 1003
    stages = []
    for k, expr in enumerate((dg.div(), dg.grad(), dg.face_mass(4))):
        host = generate_host_input_arrays(expr, E, np_seed=k)
        stages.append((expr, {n: torch.from_numpy(v).cuda() for n, v in host.items()}))
    stages[1][1]["J"], stages[1][1]["R"] = stages[0][1]["J"], stages[0][1]["R"]
    out_dicts, bufs = [], []
    for expr, _ in stages:
        shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
        n = int(np.prod(shape))
        od = {}
        for name in expr.output_names:
            buf = torch.full((n + 2 * guard,), -7.25, dtype=torch.float64, device="cuda")
            bufs.append((buf, n))
            od[name] = buf[guard:guard + n].view(shape)
        out_dicts.append(od)
    f.evaluate_operator(stages, 0, out_dicts=out_dicts, transform=variant, wait=True)
    for buf, n in bufs:
        assert bool((buf[:guard] == -7.25).all()) and bool((buf[guard + n:] == -7.25).all())
        assert not bool((buf[guard:guard + n] == -7.25).any())
