"""Parity of the HIP path with the oracle (the first gate).  Every comparison
goes product (C ABI via ctypes) vs oracle (numpy ground truth); tolerance is
north_star's 1e-12 relative (the reference itself only asks for 1e-10:
src/feinsum/measure.py:181-183)."""

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd import _hip
from feinsum_amd.measure import generate_host_input_arrays

import dg

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _hip.load_library()   # fail loudly if the extension is missing
    return torch


def _run(torch, expr, host, transform=None):
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    outs = f.evaluate(expr, 0, dev, transform=transform, wait=True)
    return {k: v.cpu().numpy() for k, v in outs.items()}


def _oracle(expr, host):
    from oracle import np_oracle

    return {name: np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in row]])[0]
            for name, row in zip(expr.output_names, expr.args)}


def _assert_close(got, ref, tol=TOL):
    from oracle import np_oracle

    assert set(got) == set(ref)
    for k in ref:
        assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype
        assert np.isfinite(got[k]).all()
        assert np_oracle.max_rel_err(got[k], ref[k]) <= tol, k
        if ref[k].size:   # elementwise too (inputs are positive: no cancellation)
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-11, atol=0)


@pytest.mark.parametrize("name", sorted(dg.GOLDEN_CASES))
@pytest.mark.parametrize("E", [1, 7, 37])
def test_golden_vectors(torch_cuda, golden_dir, name, E):
    z = np.load(golden_dir / f"{name}_E{E}.npz")
    expr = dg.GOLDEN_CASES[name]()
    host = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    _assert_close(_run(torch_cuda, expr, host), ref)


FAMILIES = {"grad": dg.grad, "div": dg.div, "face_mass": dg.face_mass,
            "face_mass_ifj_fe": dg.face_mass_ifj_fe,
            # transposed-operator siblings
            "grad_t": dg.grad_t, "div_t": dg.div_t, "face_mass_jfi_fe": dg.face_mass_jfi_fe,
            "face_mass_fji": dg.face_mass_fji}


@pytest.mark.parametrize("fam", sorted(FAMILIES))
@pytest.mark.parametrize("E", [0, 1, 15, 16, 17, 31, 32, 100, 1000, 4099, 10007])
@pytest.mark.parametrize("variant", ["auto", "generic"])
def test_families_vs_oracle(torch_cuda, fam, E, variant):
    # empty, sub-tile, exact-tile, ragged and odd element counts; both kernel variants
    expr = FAMILIES[fam]()
    host = generate_host_input_arrays(expr, E, np_seed=E + 1)
    _assert_close(_run(torch_cuda, expr, host, transform=variant), _oracle(expr, host))


@pytest.mark.parametrize("fam", ["grad", "div", "face_mass"])
def test_forced_mfma_variant(torch_cuda, fam):
    expr = FAMILIES[fam]()
    host = generate_host_input_arrays(expr, 4096, np_seed=5)
    _assert_close(_run(torch_cuda, expr, host, transform={"variant": "mfma"}), _oracle(expr, host))
    with pytest.raises(NotImplementedError):   # FE_EUNSUPPORTED: Np = 84 (p = 6) is not compiled for MFMA
        e84 = dg.grad(84)
        _run(torch_cuda, e84, generate_host_input_arrays(e84, 64), transform="mfma")


@pytest.mark.parametrize("Np", [4, 10, 20])
@pytest.mark.parametrize("E", [15, 16, 79, 80, 81, 1000, 10007])
def test_grad_lower_orders_mfma(torch_cuda, Np, E):
    # p = 1, 2, 3: the (Np, M) instantiations of the grad template; wave tiles of 80 / 48 / 32
    # elements, so E around those sizes exercises tile + generic-remainder splits
    expr = dg.grad(Np)
    host = generate_host_input_arrays(expr, E, np_seed=Np + E)
    ref = _oracle(expr, host)
    _assert_close(_run(torch_cuda, expr, host, transform="mfma"), ref)
    _assert_close(_run(torch_cuda, expr, host, transform="generic"), ref)


@pytest.mark.parametrize("Np", [4, 10, 20])
@pytest.mark.parametrize("E", [15, 16, 79, 80, 81, 1000, 10007])
def test_div_lower_orders_mfma(torch_cuda, Np, E):
    expr = dg.div(Np)
    host = generate_host_input_arrays(expr, E, np_seed=2 * Np + E)
    ref = _oracle(expr, host)
    _assert_close(_run(torch_cuda, expr, host, transform="mfma"), ref)
    _assert_close(_run(torch_cuda, expr, host, transform="generic"), ref)


@pytest.mark.parametrize("order", [(4, 3), (10, 6), (20, 10)])
@pytest.mark.parametrize("E", [15, 16, 63, 64, 65, 1000, 4099])
@pytest.mark.parametrize("b", [2, 4, 5, 7])
def test_face_mass_lower_orders_mfma(torch_cuda, order, E, b):
    # p = 1, 2, 3 tetrahedra: (Np, Nfp) = (4, 3), (10, 6), (20, 10); fields grouped by up to 4
    Np, Nfp = order
    for expr in (dg.face_mass(b, Np=Np, Nfp=Nfp), dg.face_mass_ifj_fe(b, Np=Np, Nfp=Nfp)):
        host = generate_host_input_arrays(expr, E, np_seed=Np + E + b)
        _assert_close(_run(torch_cuda, expr, host, transform="mfma"), _oracle(expr, host))


@pytest.mark.parametrize("Np", [4, 10, 20])
def test_transposed_operators_lower_orders(torch_cuda, Np):
    Nfp = {4: 3, 10: 6, 20: 10}[Np]
    for expr in (dg.grad_t(Np), dg.div_t(Np), dg.face_mass_jfi_fe(4, Np=Np, Nfp=Nfp),
                 dg.face_mass_fji(3, Np=Np, Nfp=Nfp)):
        host = generate_host_input_arrays(expr, 1003, np_seed=Np)
        ref = _oracle(expr, host)
        _assert_close(_run(torch_cuda, expr, host, transform="mfma"), ref)
        _assert_close(_run(torch_cuda, expr, host, transform="generic"), ref)


@pytest.mark.parametrize("Np", [4, 10, 20, 35, 56, 84])
@pytest.mark.parametrize("E", [31, 32, 97, 1003])
def test_div_components(torch_cuda, Np, E):
    # 'se,sij,ej->ei' x 3 (test/test_codegen.py:34-66) and its 'es' / transposed-operator layouts
    import feinsum_amd as f2

    variants = ["auto", "generic"] + (["mfma"] if Np != 84 else [])
    for subs, jshape in (("se,sij,ej->ei", (3, "E")), ("es,sij,ej->ei", ("E", 3)), ("se,sji,ej->ei", (3, "E"))):
        expr = f2.batched_einsum(subs, [[f2.array("J" + c, jshape), f2.array("R", (3, Np, Np)),
                                         f2.array("u" + c, ("E", Np))] for c in "xyz"])
        host = generate_host_input_arrays(expr, E, np_seed=Np + E)
        ref = _oracle(expr, host)
        for v in variants:
            _assert_close(_run(torch_cuda, expr, host, transform=v), ref)


@pytest.mark.parametrize("Np", [4, 10, 20, 35])
@pytest.mark.parametrize("b", [2, 3, 5, 6, 8, 11])
@pytest.mark.parametrize("E", [15, 16, 97, 1003])
def test_batched_grad_div_share_geometry_factors(torch_cuda, Np, b, E):
    # 'xre,rij,ej->xei' x b and 'xre,rij,xej->ei' x b with one J and one operator
    # (tuning/impls/batched_xre_rij_ej_to_xei.py, batched_xre_rij_xej_to_ei.py; b = 3, 5, 6 are the
    # archive's cases): all fields go through one launch, more than 8 through two
    for expr in (dg.batched_grad(b, Np), dg.batched_div(b, Np)):
        host = generate_host_input_arrays(expr, E, np_seed=Np + b + E)
        ref = _oracle(expr, host)
        _assert_close(_run(torch_cuda, expr, host, transform="mfma"), ref)
        _assert_close(_run(torch_cuda, expr, host, transform="generic"), ref)


def test_batched_grad_div_row_grouping(torch_cuda):
    # rows that do not share J (or the operator) are separate launches; transposed operator
    import feinsum_amd as f2

    E, Np = 1003, 35
    rows = [[f2.array("Ja" if i < 2 else "Jb", (3, 3, "E")), f2.array("R" if i != 3 else "R2", (3, Np, Np)),
             f2.array(f"u{i}", ("E", Np))] for i in range(5)]
    for expr in (f2.batched_einsum("xre,rij,ej->xei", rows), dg.batched_grad(4, Np, "rji"),
                 dg.batched_div(4, Np, "rji")):
        host = generate_host_input_arrays(expr, E, np_seed=5)
        ref = _oracle(expr, host)
        _assert_close(_run(torch_cuda, expr, host), ref)
        _assert_close(_run(torch_cuda, expr, host, transform="generic"), ref)


def test_batched_grad_matches_single_field_launches_bitwise(torch_cuda):
    # the multi-field walk only changes the order of tiles, never the arithmetic of a tile
    E, Np, b = 16 * 700 + 5, 35, 3
    expr = dg.batched_grad(b, Np)
    host = generate_host_input_arrays(expr, E, np_seed=1)
    got = _run(torch_cuda, expr, host)
    single = dg.grad(Np)
    for k, name in enumerate(expr.output_names):
        one = _run(torch_cuda, single, {"J": host["J"], "R": host["R"], "u": host[f"u{k}"]})
        assert np.array_equal(got[name], next(iter(one.values())))


@pytest.mark.parametrize("Np", [4, 10, 20, 35, 56])
@pytest.mark.parametrize("E", [15, 16, 97, 1003])
def test_cross_product_batch_shares_the_operator_product(torch_cuda, Np, E):
    # 're,rji,ej->ei' x 12 over six fields and three J arrays
    # (tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231): rows sharing u and D are one
    # grad-type launch with two output planes per field
    from feinsum_amd.measure import _bind

    for op in ("rji", "rij"):
        expr = dg.cross_product_batch(Np, op)
        host = generate_host_input_arrays(expr, E, np_seed=Np + E)
        ref = _oracle(expr, host)
        for v in ["auto", "generic"] + (["mfma"] if Np != 56 else []):
            _assert_close(_run(torch_cuda, expr, host, transform=v), ref)
    dev = {k: torch_cuda.from_numpy(v).cuda() for k, v in host.items()}
    _, bound, _ = _bind(expr, 0, dev, None, None)
    assert bound.group_family == 6 and len(bound.groups) == 1 and bound.groups[0].b == 6


def test_div_component_rows_that_cannot_share_stay_separate(torch_cuda):
    from feinsum_amd.measure import _bind

    E, Np = 1003, 35
    cases = {
        "three planes": {"u": ("Jx", "Jy", "Jz"), "w": ("Jx", "Jy", "Jz")},        # a grad, row by row
        "uneven": {"u": ("Jx", "Jy"), "w": ("Jz",)},
        "one each": {"ux": ("Jx",), "uy": ("Jy",), "uz": ("Jz",)},
        "four J arrays": {"u": ("Ja", "Jb"), "w": ("Jc", "Jd")},
    }
    for name, fields in cases.items():
        expr = dg.cross_product_batch(Np, "rij", fields)
        host = generate_host_input_arrays(expr, E, np_seed=3)
        _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))
        dev = {k: torch_cuda.from_numpy(v).cuda() for k, v in host.items()}
        _, bound, _ = _bind(expr, 0, dev, None, None)
        assert (bound.group_family == 6) == (name == "three planes"), name


@pytest.mark.parametrize("Np", [3, 4, 6, 10, 15, 20, 35, 56, 84])
@pytest.mark.parametrize("E", [15, 16, 127, 128, 129, 1003, 10007])
def test_element_local_operator(torch_cuda, Np, E):
    # 'e,ij,ej->ei' x b (b = 4, 5, 6, 16 in the archive) and 'ij,ej->ei', plain and transposed operator:
    # the one-component instances of the div template; wave tiles of 32 .. 128 elements
    variants = ["auto", "generic", "tiled"] + (["mfma"] if Np != 84 else [])
    for expr in (dg.mass_apply(4, Np), dg.mass_apply(5, Np, "ji"), dg.operator_apply(Np), dg.operator_apply(Np, "ji")):
        host = generate_host_input_arrays(expr, E, np_seed=Np + E)
        ref = _oracle(expr, host)
        for v in variants:
            _assert_close(_run(torch_cuda, expr, host, transform=v), ref)


def test_element_local_operator_many_fields(torch_cuda):
    expr = dg.mass_apply(16, 20)
    host = generate_host_input_arrays(expr, 4099, np_seed=16)
    _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))


@pytest.mark.parametrize("b", [1, 2, 3, 5, 8, 9, 19])
def test_face_mass_field_counts(torch_cuda, b):
    # b = 1 (generic only), odd counts, > 8 fields (several launches), 19 as in the archive
    expr = dg.face_mass(b)
    host = generate_host_input_arrays(expr, 333, np_seed=b)
    _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))


@pytest.mark.parametrize("Np", [4, 10, 20, 56])
def test_other_orders_take_the_generic_kernels(torch_cuda, Np):
    for expr in (dg.grad(Np), dg.div(Np)):
        host = generate_host_input_arrays(expr, 777, np_seed=Np)
        _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))
    expr = dg.face_mass(4, Np=Np, nf=4, Nfp=6)
    host = generate_host_input_arrays(expr, 100, np_seed=Np)
    _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))


def test_generic_einsum_kernel(torch_cuda):
    # einsums outside the DG families: test/test_codegen.py:34-66, test/test_measure.py:33-52
    expr = dg.batched_div_components()
    host = generate_host_input_arrays(expr, 300)
    _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))
    A = f.array("A", ("I", 4), "float32")
    mv = f.batched_einsum("ij, j -> i", [[A, f.array("x", 4, "float32")], [A, f.array("y", 4, "float32")]])
    host = generate_host_input_arrays(mv, 1000)
    got, ref = _run(torch_cuda, mv, host), _oracle(mv, host)
    for k in ref:
        assert got[k].dtype == np.float32
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-6, atol=1e-6)
    # an output-layout sibling of grad that is NOT the kernel family (xie instead of xei)
    g = f.einsum("xre,rij,ej->xie", f.array("J", (3, 3, "E")), f.array("R", (3, 35, 35)),
                 f.array("u", ("E", 35)))
    host = generate_host_input_arrays(g, 50)
    _assert_close(_run(torch_cuda, g, host), _oracle(g, host))


@pytest.mark.parametrize("E", [1, 7, 64, 1003, 100003])
def test_generic_einsum_reductions_and_pointwise(torch_cuda, E):
    # the remaining families of the reference's archive (tuning/impls/ij_j_to_i.py, ij_to_i.py,
    # ij_ij_to_ij.py, ijk_ijk_to_ijk.py): lane groups per output for contiguous summation
    # indices, a vectorised stream for pointwise products, the plain mapping otherwise
    cases = [
        f.einsum("ej,j->e", f.array("A", ("E", 35)), f.array("w", (35,))),
        f.einsum("ej->e", f.array("A", ("E", 35))),
        f.einsum("ej->e", f.array("A", ("E", 10))),
        f.einsum("ej->e", f.array("A", ("E", 3))),
        f.einsum("je->e", f.array("A", (35, "E"))),                           # summation index not contiguous
        f.einsum("ej,ej->ej", f.array("A", ("E", 35)), f.array("B", ("E", 35))),
        f.einsum("fej,fej->fej", f.array("A", (4, "E", 15)), f.array("B", (4, "E", 15))),
        f.einsum("ej,ej,ej->ej", f.array("A", ("E", 3)), f.array("B", ("E", 3)), f.array("C", ("E", 3))),
        f.einsum("ej,j->ej", f.array("A", ("E", 35)), f.array("w", (35,))),   # broadcast: not the stream path
        f.einsum("erj,rij->ei", f.array("A", ("E", 3, 35)), f.array("D", (3, 35, 35))),   # two summation indices
    ]
    for expr in cases:
        host = generate_host_input_arrays(expr, E, np_seed=E)
        _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))
    # a pointwise product on buffers that are only 8-byte aligned
    expr = cases[5]
    host = generate_host_input_arrays(expr, E, np_seed=1)
    torch = torch_cuda
    dev = {}
    for k, v in host.items():
        buf = torch.empty(v.size + 1, dtype=torch.float64, device="cuda")
        buf[1:] = torch.from_numpy(v).cuda().reshape(-1)
        dev[k] = buf[1:].view(v.shape)
    out = f.evaluate(expr, 0, dev, wait=True)
    _assert_close({k: v.cpu().numpy() for k, v in out.items()}, _oracle(expr, host))


@pytest.mark.parametrize("Np,Nfp", [(3, 2), (6, 3), (10, 4), (15, 5), (21, 6), (28, 7)])
@pytest.mark.parametrize("E", [5, 127, 128, 129, 1003, 6007])
def test_two_dimensional_operators(torch_cuda, Np, Nfp, E):
    # triangles (ndim = 2, three faces): grad / div through fe_grad_f64 / fe_div_f64 -- MFMA instances
    # of the div template for p = 1..5 (grad by components), the tiled kernel otherwise (p = 6) -- and
    # the lift on the face-mass entry point with nf = 3
    import feinsum_amd as f2

    grad2 = f2.einsum("xre,rij,ej->xei", f2.array("J", (2, 2, "E")), f2.array("R", (2, Np, Np)), f2.array("u", ("E", Np)))
    div2 = f2.einsum("xre,rij,xej->ei", f2.array("J", (2, 2, "E")), f2.array("R", (2, Np, Np)),
                     f2.array("u", (2, "E", Np)))
    lift2 = f2.batched_einsum("ef,fij,fej->ei", [[f2.array("J", ("E", 3)), f2.array("R", (3, Np, Nfp)),
                                                  f2.array(f"v{k}", (3, "E", Nfp))] for k in range(3)])
    bgrad2 = f2.batched_einsum("xre,rji,ej->xei", [[f2.array("J", (2, 2, "E")), f2.array("R", (2, Np, Np)),
                                                    f2.array(f"u{k}", ("E", Np))] for k in range(3)])
    # div components of triangles ('se,sij,ej->ei' with two components; tuning/impls/re_rij_ej_to_ei.py with ndim = 2),
    # in both J layouts and with the transposed operator
    comp2 = f2.batched_einsum("se,sij,ej->ei", [[f2.array("J" + c, (2, "E")), f2.array("R", (2, Np, Np)),
                                                f2.array("u" + c, ("E", Np))] for c in "xy"])
    comp2_es = f2.batched_einsum("es,sji,ej->ei", [[f2.array("J" + c, ("E", 2)), f2.array("R", (2, Np, Np)),
                                                   f2.array("u" + c, ("E", Np))] for c in "xy"])
    for expr in (grad2, div2, lift2, bgrad2, comp2, comp2_es):
        assert f2.match_family(expr) is not None
        host = generate_host_input_arrays(expr, E, np_seed=Np)
        ref = _oracle(expr, host)
        _assert_close(_run(torch_cuda, expr, host), ref)
        _assert_close(_run(torch_cuda, expr, host, transform="tiled"), ref)
        if Np <= 21:
            _assert_close(_run(torch_cuda, expr, host, transform="mfma"), ref)
    # the lift in the 'ifj,fe' spelling and with other field counts
    for b in (2, 5):
        lift = f2.batched_einsum("ifj,fe,fej->ei", [[f2.array("L", (Np, 3, Nfp)), f2.array("J", (3, "E")),
                                                     f2.array(f"v{k}", (3, "E", Nfp))] for k in range(b)])
        host = generate_host_input_arrays(lift, E, np_seed=b)
        _assert_close(_run(torch_cuda, lift, host), _oracle(lift, host))


@pytest.mark.parametrize("E", [1, 15, 16, 17, 63, 64, 65, 1003, 5000])
def test_p5_on_the_matrix_cores(torch_cuda, E):
    # Np = 56: the A fragments in LDS, one block per CU -- grad by components, div with its u planes
    # streamed through two buffers, face-mass; plain, transposed, batched
    for expr in (dg.grad(56), dg.grad_t(56), dg.batched_grad(3, 56), dg.div(56), dg.div_t(56), dg.batched_div(3, 56),
                 dg.face_mass(4, Np=56, Nfp=21), dg.face_mass_jfi_fe(3, Np=56, Nfp=21)):
        host = generate_host_input_arrays(expr, E, np_seed=E)
        ref = _oracle(expr, host)
        for v in ("auto", "mfma", "tiled", "generic"):
            _assert_close(_run(torch_cuda, expr, host, transform=v), ref)


@pytest.mark.parametrize("Np,Nfp", [(4, 3), (20, 10), (35, 15), (56, 21)])
@pytest.mark.parametrize("E", [1, 63, 64, 65, 1003, 5000])
def test_tiled_kernel_all_families(torch_cuda, Np, Nfp, E):
    # the LDS-tiled VALU kernel at compiled orders (forced), at p = 5 and p = 6 (what AUTO picks
    # there), partial tiles included
    exprs = [dg.grad(Np), dg.div(Np), dg.grad_t(Np), dg.div_t(Np), dg.batched_grad(3, Np), dg.batched_div(2, Np),
             dg.face_mass(4, Np=Np, Nfp=Nfp), dg.face_mass_ifj_fe(3, Np=Np, Nfp=Nfp),
             dg.face_mass_jfi_fe(2, Np=Np, Nfp=Nfp), dg.face_mass_fji(9, Np=Np, Nfp=Nfp),
             dg.mass_apply(4, Np), dg.operator_apply(Np, "ji"), dg.batched_div_components(Np)]
    for expr in exprs:
        host = generate_host_input_arrays(expr, E, np_seed=Np + E)
        ref = _oracle(expr, host)
        _assert_close(_run(torch_cuda, expr, host, transform="tiled"), ref)
        if Np > 35:
            _assert_close(_run(torch_cuda, expr, host), ref)


def test_tiled_kernel_that_does_not_fit_falls_back(torch_cuda):
    # 3 x 150 x 150 doubles = 540 KB: no room in LDS; AUTO takes the generic kernel, "tiled" refuses
    expr = dg.grad(150)
    host = generate_host_input_arrays(expr, 70, np_seed=1)
    _assert_close(_run(torch_cuda, expr, host), _oracle(expr, host))
    with pytest.raises(NotImplementedError, match="LDS"):
        _run(torch_cuda, expr, host, transform="tiled")


def test_outputs_are_overwritten_not_accumulated(torch_cuda):
    torch = torch_cuda
    expr = dg.grad()
    host = generate_host_input_arrays(expr, 100)
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    out = torch.full((3, 100, 35), float("nan"), dtype=torch.float64, device="cuda")
    res = f.evaluate(expr, 0, dev, out_dict={"_fe_out": out}, wait=True)
    assert res["_fe_out"].data_ptr() == out.data_ptr()
    _assert_close({"_fe_out": out.cpu().numpy()}, _oracle(expr, host))


# ---- BASELINE.json full size (E = 1e6): size-independent properties ---------------------

@pytest.fixture(scope="module")
def big_grad(torch_cuda):
    torch = torch_cuda
    E = 1_000_000
    g = torch.Generator(device="cuda").manual_seed(0)
    J = torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g)
    D = torch.rand((3, 35, 35), dtype=torch.float64, device="cuda", generator=g)
    u = torch.rand((E, 35), dtype=torch.float64, device="cuda", generator=g)
    out = f.evaluate(dg.grad(), 0, {"J": J, "R": D, "u": u}, wait=True)["_fe_out"]
    return E, J, D, u, out


def test_full_size_mfma_vs_generic_and_sampled_oracle(torch_cuda, big_grad):
    torch = torch_cuda
    from oracle import np_oracle

    E, J, D, u, out = big_grad
    gen = f.evaluate(dg.grad(), 0, {"J": J, "R": D, "u": u}, transform="generic", wait=True)["_fe_out"]
    err = float((out - gen).abs().max() / gen.abs().max())
    assert err <= TOL
    # oracle on first / middle / last 1000 elements
    for s in (0, E // 2 - 500, E - 1000):
        sl = slice(s, s + 1000)
        ref = np_oracle.reference_outputs("xre,rij,ej->xei", [[J[:, :, sl].cpu().numpy(), D.cpu().numpy(),
                                                               u[sl].cpu().numpy()]])[0]
        assert np_oracle.max_rel_err(out[:, sl].cpu().numpy(), ref) <= TOL


def test_full_size_linearity_and_locality(torch_cuda, big_grad):
    torch = torch_cuda
    E, J, D, u, out = big_grad
    expr = dg.grad()
    # exact linearity under power-of-two scaling (bit-exact in binary floating point)
    out2 = f.evaluate(expr, 0, {"J": J, "R": D, "u": u * 4.0}, wait=True)["_fe_out"]
    assert torch.equal(out2, out * 4.0)
    # element locality: a tile-aligned sub-batch evaluated alone is bit-identical
    s, n = 16 * 1234, 16 * 500
    sub = f.evaluate(expr, 0, {"J": J[:, :, s:s + n].contiguous(), "R": D, "u": u[s:s + n].contiguous()},
                     wait=True)["_fe_out"]
    assert torch.equal(sub, out[:, s:s + n])
    # permuting whole elements permutes the output (bit-exact: per-element arithmetic order
    # does not depend on the element's position)
    perm = torch.randperm(E, device="cuda")
    outp = f.evaluate(expr, 0, {"J": J[:, :, perm].contiguous(), "R": D, "u": u[perm].contiguous()},
                      wait=True)["_fe_out"]
    assert torch.equal(outp, out[:, perm])
    # determinism
    again = f.evaluate(expr, 0, {"J": J, "R": D, "u": u}, wait=True)["_fe_out"]
    assert torch.equal(again, out)


@pytest.mark.parametrize("fam", ["div", "face_mass"])
def test_full_size_div_and_face_mass(torch_cuda, fam):
    torch = torch_cuda
    from oracle import np_oracle

    E = 1_000_000
    expr = FAMILIES[fam]()
    g = torch.Generator(device="cuda").manual_seed(1)
    dev = {}
    for name in sorted(expr.all_args):
        shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[name])
        dev[name] = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
    outs = f.evaluate(expr, 0, dev, wait=True)
    gens = f.evaluate(expr, 0, dev, transform="generic", wait=True)
    for k in outs:
        assert float((outs[k] - gens[k]).abs().max() / gens[k].abs().max()) <= TOL
    # sampled oracle on the last 500 elements (includes no ragged tail: E % 16 == 0)
    sl = slice(E - 500, E)
    host = {}
    for name, t in dev.items():
        shape = expr.arg_to_shape[name]
        idx = tuple(sl if isinstance(d, f.SizeParam) else slice(None) for d in shape)
        host[name] = t[idx].cpu().numpy()
    ref = _oracle(expr, host)
    for k in ref:
        assert np_oracle.max_rel_err(outs[k][sl].cpu().numpy(), ref[k]) <= TOL
    # exact linearity in the field operand
    scaled = dict(dev)
    vname = "u" if fam == "div" else "v0"
    scaled[vname] = dev[vname] * 2.0
    outs2 = f.evaluate(expr, 0, scaled, wait=True)
    assert torch.equal(outs2["_fe_out"], outs["_fe_out"] * 2.0)
    # position independence (as for grad in test_full_size_linearity_and_locality): a tile-aligned sub-batch evaluated
    # alone, and the whole batch with its elements permuted, give the same bits per element -- a tile's arithmetic does
    # not depend on where it falls in a wave's walk, hence not on E or the CU count
    def restricted(index):
        sub = {}
        for name, t in dev.items():
            idx = tuple(index if isinstance(d, f.SizeParam) else slice(None) for d in expr.arg_to_shape[name])
            sub[name] = t[idx].contiguous()
        return sub

    s0, n = 16 * 4321, 16 * 700
    part = f.evaluate(expr, 0, restricted(slice(s0, s0 + n)), wait=True)
    for k in outs:
        assert torch.equal(part[k], outs[k][s0:s0 + n]), k
    perm = torch.randperm(E, device="cuda")
    permuted = f.evaluate(expr, 0, restricted(perm), wait=True)
    for k in outs:
        assert torch.equal(permuted[k], outs[k][perm]), k
    again = f.evaluate(expr, 0, dev, wait=True)
    for k in outs:
        assert torch.equal(again[k], outs[k]), k


FULL_SIZE_SIBLINGS = {
    "batched_grad_b3": lambda: dg.batched_grad(3),
    "batched_div_b2": lambda: dg.batched_div(2),
    "cross_product": lambda: dg.cross_product_batch(),
    "mass_apply_b4": lambda: dg.mass_apply(4),
    "operator_apply": lambda: dg.operator_apply(),
    "div_components": lambda: dg.batched_div_components(),
}


@pytest.mark.parametrize("name", sorted(FULL_SIZE_SIBLINGS))
def test_full_size_siblings(torch_cuda, name):
    # the multi-field / planes / one-component launches at E = 1e6 (+ a ragged tail): MFMA against
    # the generic kernels everywhere, the oracle on the first and the last 300 elements
    torch = torch_cuda
    from oracle import np_oracle

    E = 1_000_000 + 7
    expr = FULL_SIZE_SIBLINGS[name]()
    g = torch.Generator(device="cuda").manual_seed(2)
    dev = {}
    for arg in sorted(expr.all_args):
        shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[arg])
        dev[arg] = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
    outs = f.evaluate(expr, 0, dev, wait=True)
    gens = f.evaluate(expr, 0, dev, transform="generic", wait=True)
    for k in outs:
        assert float((outs[k] - gens[k]).abs().max() / gens[k].abs().max()) <= TOL
    del gens
    for sl in (slice(0, 300), slice(E - 300, E)):
        host = {}
        for arg, t in dev.items():
            idx = tuple(sl if isinstance(d, f.SizeParam) else slice(None) for d in expr.arg_to_shape[arg])
            host[arg] = t[idx].cpu().numpy()
        ref = _oracle(expr, host)
        long_axis = [isinstance(d, f.SizeParam) for d in expr.shape].index(True)
        for k in ref:
            got = outs[k][(slice(None),) * long_axis + (sl,)].cpu().numpy()
            assert np_oracle.max_rel_err(got, ref[k]) <= TOL


def test_full_size_wave_operator(torch_cuda):
    # one launch for div + grad + lift at E = 1e6 + 9 is bitwise the three separate launches
    torch = torch_cuda
    E = 1_000_000 + 9
    exprs = [dg.div(), dg.grad(), dg.face_mass(4)]
    g = torch.Generator(device="cuda").manual_seed(3)
    devs = []
    for expr in exprs:
        dev = {}
        for arg in sorted(expr.all_args):
            shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[arg])
            dev[arg] = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
        devs.append(dev)
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    stages = list(zip(exprs, devs))
    assert f.bind_operator(stages, 0).entry_points == ("fe_waveop3d_f64",)
    fused = f.evaluate_operator(stages, 0, wait=True)
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)
    for a, b in zip(fused, plain):
        for k in a:
            assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all())


# ---- sizes at which every wave walks several tiles, and BASELINE config 5's element count ----

def _device_inputs(torch, expr, E, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dev = {}
    for arg in sorted(expr.all_args):
        shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[arg])
        dev[arg] = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
    return dev


def _sampled_oracle_check(expr, dev, outs, E, slices):
    from oracle import np_oracle

    long_axis = [isinstance(d, f.SizeParam) for d in expr.shape].index(True)
    for sl in slices:
        host = {}
        for arg, t in dev.items():
            idx = tuple(sl if isinstance(d, f.SizeParam) else slice(None) for d in expr.arg_to_shape[arg])
            host[arg] = t[idx].cpu().numpy()
        ref = _oracle(expr, host)
        for k in ref:
            got = outs[k][(slice(None),) * long_axis + (sl,)].cpu().numpy()
            assert np_oracle.max_rel_err(got, ref[k]) <= TOL, (k, sl)


P5_MULTI_TILE = {
    "grad": lambda: dg.grad(56), "grad_t": lambda: dg.grad_t(56), "batched_grad_b2": lambda: dg.batched_grad(2, 56),
    "div": lambda: dg.div(56), "batched_div_b2": lambda: dg.batched_div(2, 56),
    "div_components": lambda: dg.batched_div_components(56),
    "face_mass_b4": lambda: dg.face_mass(4, Np=56, Nfp=21), "face_mass_b3_jfi": lambda: dg.face_mass_jfi_fe(3, Np=56, Nfp=21),
    "cross_product": lambda: dg.cross_product_batch(56),
}


@pytest.mark.parametrize("name", sorted(P5_MULTI_TILE))
def test_p5_every_wave_walks_several_tiles(torch_cuda, name):
    """Np = 56 at E = 70 003: more than 16 x 8 x 256 elements, so the eight-wave blocks (one per CU)
    loop to a second and third tile -- the path where a wave's input buffer doubles as its output
    transposition buffer and the next tile (with a new J tile) is requested behind the stores.
    MFMA against the generic kernels over the whole array, the oracle on head / middle / tail."""
    torch = torch_cuda
    E = 70_003
    expr = P5_MULTI_TILE[name]()
    dev = _device_inputs(torch, expr, E, seed=11)
    outs = f.evaluate(expr, 0, dev, transform="mfma", wait=True)
    gens = f.evaluate(expr, 0, dev, transform="generic", wait=True)
    for k in outs:
        assert bool(torch.isfinite(outs[k]).all())
        assert float((outs[k] - gens[k]).abs().max() / gens[k].abs().max()) <= TOL, k
    auto = f.evaluate(expr, 0, dev, wait=True)
    for k in outs:
        assert torch.equal(auto[k], outs[k])            # AUTO picks the MFMA kernels at p = 5
    _sampled_oracle_check(expr, dev, outs, E, (slice(0, 200), slice(E // 2 - 100, E // 2 + 100), slice(E - 200, E)))


def test_full_size_graddiv_single_launch(torch_cuda):
    """BASELINE config 3 at its own size: div + grad sharing J and D in one launch at E = 1e6 + 5 is
    bitwise the two separate launches; sampled oracle on both outputs."""
    torch = torch_cuda
    E = 1_000_000 + 5
    exprs = [dg.div(), dg.grad()]
    devs = [_device_inputs(torch, e, E, seed=21 + k) for k, e in enumerate(exprs)]
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    stages = list(zip(exprs, devs))
    assert f.bind_operator(stages, 0).entry_points == ("fe_graddiv3d_f64",)
    fused = f.evaluate_operator(stages, 0, wait=True)
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)
    for a, b in zip(fused, plain):
        for k in a:
            assert torch.equal(a[k], b[k])
    for expr, dev, out in zip(exprs, devs, fused):
        _sampled_oracle_check(expr, dev, out, E, (slice(0, 300), slice(E // 2, E // 2 + 300), slice(E - 300, E)))


def test_config5_eight_million_elements(torch_cuda):
    """BASELINE config 5's whole element count on ONE GPU: E = 8e6 (grad out alone is 6.7 GB; byte
    offsets pass 2^32 in every array).  Plain grad and the div + grad + lift pipeline in one launch:
    sampled oracle on first / middle / last slices, and the single launch bitwise equal to the three
    separate launches.  (Sharded over 8 ranks each rank runs the 1e6 case of the tests above.)"""
    torch = torch_cuda
    E = 8_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~65 GB of device memory")
    exprs = [dg.div(), dg.grad(), dg.face_mass(4)]
    devs = [_device_inputs(torch, e, E, seed=31 + k) for k, e in enumerate(exprs)]
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    slices = (slice(0, 200), slice(E // 2 - 100, E // 2 + 100), slice(5_000_001, 5_000_201), slice(E - 200, E))
    # plain grad
    grad_out = f.evaluate(exprs[1], 0, devs[1], wait=True)
    _sampled_oracle_check(exprs[1], devs[1], grad_out, E, slices)
    # the pipeline as one launch
    stages = list(zip(exprs, devs))
    assert f.bind_operator(stages, 0).entry_points == ("fe_waveop3d_f64",)
    fused = f.evaluate_operator(stages, 0, wait=True)
    assert torch.equal(fused[1]["_fe_out"], grad_out["_fe_out"])
    del grad_out
    for expr, dev, out in zip(exprs, devs, fused):
        _sampled_oracle_check(expr, dev, out, E, slices)
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)
    for a, b in zip(fused, plain):
        for k in a:
            assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("Np", [4, 10, 20, 35])
@pytest.mark.parametrize("E", [16, 17, 79, 1000, 32784, 70003])
def test_div_split_walk_matches_the_mfma_kernel(torch_cuda, Np, E):
    """FE_VARIANT_MFMA_SPLIT ("mfma_split"): div walking both halves of the element range at once -- the same tiles, the
    same arithmetic, another order (odd and even tile counts, a remainder behind the last tile, several tiles per wave
    at E = 70 003), single and batched.  BITWISE equal to the plain walk: the VALU contractions are explicit fused
    multiply-adds (round 3), so a tile's bits do not depend on its place in a wave's walk (round 2 had to relax this to
    rtol 1e-14: the compiler contracted a wave's first tile differently from its later ones)."""
    torch = torch_cuda
    for expr in (dg.div(Np), dg.batched_div(3, Np)):
        dev = _device_inputs(torch, expr, E, seed=E + Np)
        plain = f.evaluate(expr, 0, dev, transform="mfma", wait=True)
        split = f.evaluate(expr, 0, dev, transform="mfma_split", wait=True)
        generic = f.evaluate(expr, 0, dev, transform="generic", wait=True)
        assert set(plain) == set(split)
        for name in plain:
            assert torch.equal(plain[name], split[name]), (name, Np, E)
            assert torch.allclose(generic[name], split[name], rtol=1e-12, atol=0.0), (name, Np, E)


@pytest.mark.gpu
def test_split_walk_is_a_div_variant_only(torch_cuda):
    torch = torch_cuda
    for expr in (dg.grad(35), dg.face_mass(4), dg.div(56)):
        dev = _device_inputs(torch, expr, 100, seed=1)
        with pytest.raises(NotImplementedError):
            f.evaluate(expr, 0, dev, transform="mfma_split", wait=True)


# ---- float32 operands (round 3): the DG families on the LDS-tiled kernel in float --------------------

def _f32(expr):
    """The same einsum with every operand in float32."""
    rows = [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args]
    return f.batched_einsum(expr.get_subscripts(), rows)


F32_CASES = {
    "grad": lambda: dg.grad(), "grad_t": lambda: dg.grad_t(), "div": lambda: dg.div(), "face_mass": lambda: dg.face_mass(4),
    "face_mass_ifj_fe": lambda: dg.face_mass_ifj_fe(4), "div_components": lambda: dg.batched_div_components(),
    "batched_grad_b3": lambda: dg.batched_grad(3), "batched_div_b2": lambda: dg.batched_div(2),
    "mass_apply_b4": lambda: dg.mass_apply(4), "operator_apply": lambda: dg.operator_apply(),
    "grad_p3": lambda: dg.grad(20), "div_p2": lambda: dg.div(10), "grad_p5": lambda: dg.grad(56),
    # round 4: the float32 MFMA grad kernel is templated on Np (p = 1 ... 3 ran on the tiled kernel before)
    "grad_p1": lambda: dg.grad(4), "grad_p2": lambda: dg.grad(10), "grad_t_p3": lambda: dg.grad_t(20), "batched_grad_b3_p2": lambda: dg.batched_grad(3, 10),
    # round 5: the float32 MFMA div kernel over the geometry (Np, M): p = 1 ... 3
    "face_mass_p1": lambda: dg.face_mass(4, Np=4, Nfp=3), "face_mass_p2_b3": lambda: dg.face_mass(3, Np=10, Nfp=6), "face_mass_p3": lambda: dg.face_mass(4, Np=20, Nfp=10),
    "face_mass_p3_ifj_fe": lambda: dg.face_mass_ifj_fe(2, Np=20, Nfp=10), "face_mass_p2_b9": lambda: dg.face_mass(9, Np=10, Nfp=6),
    "div_p1": lambda: dg.div(4), "div_p3": lambda: dg.div(20), "div_t_p2": lambda: dg.div_t(10), "batched_div_b2_p3": lambda: dg.batched_div(2, 20),
    "face_mass_p5": lambda: dg.face_mass(4, Np=56, Nfp=21), "face_mass_b9": lambda: dg.face_mass(9),
    "div_t": lambda: dg.div_t(), "face_mass_b1": lambda: dg.face_mass(1), "face_mass_b2": lambda: dg.face_mass(2),
    "face_mass_b3": lambda: dg.face_mass(3), "face_mass_jfi_fe": lambda: dg.face_mass_jfi_fe(4), "face_mass_fji": lambda: dg.face_mass_fji(2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(F32_CASES))
@pytest.mark.parametrize("E", [1, 37, 132, 1000, 10007, 70004])
def test_float32_families(torch_cuda, name, E):
    """All-float32 DG einsums (the reference validates float32 at 1e-6: src/feinsum/measure.py:178-192) run on the
    matrix cores (grad / div / face-mass at p = 4 with E a multiple of 4: fe_grad_f32.h, fe_div_f32.h, fe_facemass_f32.h;
    E = 1000 has 62 tiles and 8 elements behind them, at E = 70 004 every wave walks several tiles and, with b fields,
    several (tile, field) units) or on the tiled kernel in float (fe_launch_f32), not on the one-thread-per-entry
    generic einsum kernel.  Compared with the
    float64 evaluation of the SAME float32 inputs: float32 rounding of a 105-term sum of values in [0, 1) allows ~1e-6
    of the largest entry."""
    torch = torch_cuda
    from feinsum_amd.family import match_family
    from feinsum_amd.measure import generate_host_input_arrays

    expr = _f32(F32_CASES[name]())
    plan = match_family(expr)
    assert plan is not None and plan.params.get("f32") == 1
    host = generate_host_input_arrays(expr, E, np_seed=E)
    assert all(a.dtype == np.float32 for a in host.values())
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    outs = f.evaluate(expr, 0, dev, wait=True)
    for out_name, row in zip(expr.output_names, expr.args):
        got = outs[out_name]
        assert got.dtype == torch.float32
        ref = np.einsum(expr.get_subscripts(), *[host[a.name].astype(np.float64) for a in row], optimize="optimal")
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= 2e-6, (name, E, err)


@pytest.mark.gpu
def test_float32_validation_and_timing(torch_cuda):
    """timeit on a float32 einsum: the reference's validation at E = 100 (atol = rtol = 1e-6) passes and the launch is
    timed through fe_time_launches(family | FE_FAMILY_F32); float32 moves half the bytes of float64."""
    from feinsum_amd import measure

    e32, e64 = _f32(dg.grad()), dg.grad()
    measure.validate_batched_einsum_transform(e32, 0, None)
    t32 = measure.timeit_details(e32, cq=0, long_dim_length=200_000, min_secs=0.2)
    gen = measure.timeit_details(e32, cq=0, long_dim_length=200_000, min_secs=0.2, transform="generic") if False else None
    assert 0 < t32.seconds_device < 5e-3 and gen is None
    assert measure._get_footprint_gbytes(e32, 200_000) == pytest.approx(0.5 * measure._get_footprint_gbytes(e64, 200_000))
    rate = f.measure_giga_op_rate(e32, cq=0, long_dim_length=200_000)
    assert set(rate) == {np.dtype("float32")} and rate[np.dtype("float32")] > 2000      # (the generic kernel: ~1000)


# ---- dynamic walk (round 3): tiles by tickets behind two static rounds ---------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["grad", "div", "face_mass", "face_mass_b3", "batched_div_b2", "batched_div_b3", "batched_grad_b2", "batched_grad_b3"])
def test_dynamic_walk_gives_the_bits_of_the_static_walk(torch_cuda, name):
    """Launches of five or more rounds hand their tiles to the waves by tickets (feinsum_amd/csrc/fe_common.h: dynamic walk):
    which wave computes a tile depends on timing, the result does not -- bitwise the static walk's at sizes around the
    switch (E = 163 840 is five rounds of 2048 waves), with tiles left over, with a remainder behind the last tile, and in
    launches of alternating sizes one after the other (a launch must leave its ticket counters zeroed: a stale count would
    skip tiles of the next one)."""
    torch = torch_cuda
    from feinsum_amd import _hip

    expr = {"grad": dg.grad, "div": dg.div, "face_mass": lambda: dg.face_mass(4), "face_mass_b3": lambda: dg.face_mass(3),
            "batched_div_b2": lambda: dg.batched_div(2), "batched_div_b3": lambda: dg.batched_div(3),
            "batched_grad_b2": lambda: dg.batched_grad(2), "batched_grad_b3": lambda: dg.batched_grad(3)}[name]()
    sizes = [163_840, 163_856, 200_003, 700_001, 163_840, 1_000_000, 180_000, 155_003, 147_456, 147_440]   # (the last three: four and a half rounds)
    before = _hip.set_tail_rounds(1 << 20)
    try:
        for E in sizes:
            dev = _device_inputs(torch, expr, E, seed=E % 1000)
            _hip.set_tail_rounds(-1)
            static = {k: v.clone() for k, v in f.evaluate(expr, 0, dev, wait=True).items()}
            for rounds in (1 << 20, 3):
                _hip.set_tail_rounds(rounds)
                for _ in range(3):
                    dynamic = f.evaluate(expr, 0, dev, wait=True)
                    for out_name in static:
                        assert torch.equal(static[out_name], dynamic[out_name]), (name, E, rounds, out_name)
    finally:
        _hip.set_tail_rounds(before)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["grad", "grad_t", "grad_p3", "grad_p2", "grad_p1", "batched_grad_b3", "div", "face_mass"])
def test_float32_dynamic_walk_gives_the_bits_of_the_static_walk(torch_cuda, name):
    """The float32 matrix-core kernels take their tiles by tickets too (round 4: fe_grad_f32.h; div and face-mass keep the
    static walk until they have one and must simply stay unaffected by the switch): bitwise the static walk's results at sizes
    of five and more rounds for every wave tile (16 M elements, M = 2 ... 8), with elements behind the last tile, in launches of
    alternating sizes, and with plain or non-temporal loads of the streamed operand."""
    torch = torch_cuda
    from feinsum_amd import _hip
    from feinsum_amd.measure import generate_host_input_arrays

    expr = _f32(F32_CASES[name]())
    sizes = [2048 * 16 * 8 * 5, 2048 * 16 * 8 * 5 + 76, 1_500_008, 400_000, 2048 * 16 * 8 * 5]
    before = _hip.set_tail_rounds(1 << 20)
    before_mib = _hip.set_temporal_loads_mib(248)
    try:
        for E in sizes:
            host = generate_host_input_arrays(expr, E, np_seed=E % 1000)
            dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
            _hip.set_tail_rounds(-1)
            _hip.set_temporal_loads_mib(0)
            static = {k: v.clone() for k, v in f.evaluate(expr, 0, dev, wait=True).items()}
            for rounds, mib in ((1 << 20, 248), (3, 248), (1 << 20, 0), (1 << 20, 1 << 20)):
                _hip.set_tail_rounds(rounds)
                _hip.set_temporal_loads_mib(mib)
                for _ in range(2):
                    dynamic = f.evaluate(expr, 0, dev, wait=True)
                    for out_name in static:
                        assert torch.equal(static[out_name], dynamic[out_name]), (name, E, rounds, mib, out_name)
            del dev, static, dynamic, host
    finally:
        _hip.set_tail_rounds(before)
        _hip.set_temporal_loads_mib(before_mib)
    assert _hip.tail_check()["dirty_words"] == 0


def _two_dimensional(name, Np, Nfp):
    J, R = f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np))
    if name == "grad2":
        return f.einsum("xre,rij,ej->xei", J, R, f.array("u", ("E", Np)))
    if name == "div2":
        return f.einsum("xre,rij,xej->ei", J, R, f.array("u", (2, "E", Np)))
    if name == "bgrad2":
        return f.batched_einsum("xre,rij,ej->xei", [[J, R, f.array(f"u{k}", ("E", Np))] for k in range(3)])
    return f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)), f.array(f"v{k}", (3, "E", Nfp))]
                                               for k in range(3)])


@pytest.mark.gpu
@pytest.mark.parametrize("name,Np,Nfp", [("grad2", 15, 5), ("div2", 15, 5), ("lift2", 15, 5), ("bgrad2", 15, 5), ("grad2", 6, 3),
                                         ("div2", 21, 6), ("lift2", 10, 4), ("batched_grad_p3", 20, 0), ("batched_div_p3", 20, 0),
                                         ("batched_grad_p2", 10, 0), ("batched_div_p1", 4, 0), ("face_mass_b5", 35, 15),
                                         ("face_mass_b3_p2", 10, 6), ("div_p5", 56, 0), ("grad_p5", 56, 0)])
def test_dynamic_walk_of_the_other_kernels(torch_cuda, name, Np, Nfp):
    """Round 4: the triangles' kernels (grad by components, div, lift), the batched launches of the orders p = 1 ... 3 and
    face-mass launches of any three or more fields take their tiles by tickets too.  Same property as
    above: the bits of the static walk, at sizes around the switch and in launches of alternating sizes one after the other."""
    torch = torch_cuda
    from feinsum_amd import _hip

    if name.startswith("batched_grad"):
        expr = dg.batched_grad(3, Np)
    elif name.startswith("batched_div"):
        expr = dg.batched_div(3, Np)
    elif name == "div_p5":
        expr = dg.div(56)           # eight waves per block, planes streamed (round 4: with tickets)
    elif name == "grad_p5":
        expr = dg.grad(56)
    elif name == "face_mass_b5":
        expr = dg.face_mass(5)
    elif name == "face_mass_b3_p2":
        expr = dg.face_mass(3, 10, 4, 6)
    else:
        expr = _two_dimensional(name, Np, Nfp)
    assert f.match_family(expr) is not None
    # a wave tile is 16 M elements (M = 1 ... 8 by kernel and order): sizes that give 5+ rounds of 2048 waves for every M
    sizes = [2048 * 16 * 8 * 5, 2048 * 16 * 8 * 5 + 77, 1_500_007, 2048 * 16 * 8 * 5]
    before = _hip.set_tail_rounds(1 << 20)
    try:
        for E in sizes:
            dev = _device_inputs(torch, expr, E, seed=E % 1000)
            _hip.set_tail_rounds(-1)
            static = {k: v.clone() for k, v in f.evaluate(expr, 0, dev, transform="mfma", wait=True).items()}
            for rounds in (1 << 20, 3):
                _hip.set_tail_rounds(rounds)
                for _ in range(2):
                    dynamic = f.evaluate(expr, 0, dev, transform="mfma", wait=True)
                    for out_name in static:
                        assert torch.equal(static[out_name], dynamic[out_name]), (name, E, rounds, out_name)
            del dev, static, dynamic
    finally:
        _hip.set_tail_rounds(before)
    assert _hip.tail_check()["dirty_words"] == 0


@pytest.mark.parametrize("name", ["div", "div_t", "batched_div3"])
@pytest.mark.parametrize("E", [16, 100, 1000, 4099, 10007, 98304, 100000, 131072, 300007])
def test_div_with_the_interleaved_b_build(torch_cuda, name, E):
    """Round 5: div launches of up to 6e5 elements run on the kernel that builds its B fragments k-quad by k-quad between the
    MFMA groups of the same wave (fe_div.h, kIlv; the default since round 5: fe_set_div_interleave).  Against the oracle at
    the sizes the reference's own regime covers (E = 98 304, 100 000, 131 072: src/feinsum/measure.py:202) and BITWISE the plain
    kernel -- the arithmetic of a tile is the same instruction sequence, only its place in the stream differs -- under the
    static walk (up to 131 072) and the dynamic one (300 007: ragged too), single and batched."""
    torch = torch_cuda
    expr = {"div": dg.div, "div_t": dg.div_t, "batched_div3": lambda: dg.batched_div(3)}[name]()
    host = generate_host_input_arrays(expr, E, np_seed=E + 3)
    before = _hip.set_div_interleave(1 << 40)
    try:
        got = _run(torch, expr, host)
        info = _hip.last_launch_info()
        _hip.set_div_interleave(0)
        plain = _run(torch, expr, host)
        info_plain = _hip.last_launch_info()
    finally:
        _hip.set_div_interleave(before)
    assert before == 37500                                    # the default: launches of up to 6e5 elements
    if name != "batched_div3" or E <= 131072:                 # (batched launches take it under the static walk only)
        assert info["interleaved"] and not info_plain["interleaved"], (info, info_plain)
    # E = 100 000: 6250 tiles = three rounds of 2048 waves + 106 -- at most an eighth of a round: those run as 424 quarter tiles of
    # four elements on v_mfma_f64_4x4x4_4b (fe_div.h, kOpQuarterTail); one field only
    assert info.get("quarter_tail") == (E == 100000 and name != "batched_div3"), info
    for k in got:
        assert np.array_equal(got[k], plain[k]), (name, E, k)
    if E <= 10007:
        _assert_close(got, _oracle(expr, host))
    else:                                                      # sampled oracle slices at the large sizes
        rng = np.random.default_rng(E)
        idx = np.sort(rng.choice(E, size=256, replace=False))
        sub = {}
        for a in sorted(expr.all_args):
            shape = expr.arg_to_shape[a]
            ax = [i for i, d in enumerate(shape) if isinstance(d, f.SizeParam)]
            sub[a] = np.take(host[a], idx, axis=ax[0]) if ax else host[a]
        ref = _oracle(expr, sub)
        for k in got:
            ax = [i for i, d in enumerate(expr.shape) if isinstance(d, f.SizeParam)][0]
            _assert_close({k: np.ascontiguousarray(np.take(got[k], idx, axis=ax))}, {k: ref[k]})


@pytest.mark.parametrize("name", ["grad", "grad_t"])
@pytest.mark.parametrize("E", [100000, 100007, 33000, 68736, 36864 + 64])
def test_grad_with_a_quarter_tile_tail(torch_cuda, name, E):
    """Round 5: a short grad launch (one field, static walk) whose last round is at most an eighth full runs that round's tiles as
    quarter tiles of four elements (fe_grad.h, kOpQuarterTail; fe_set_grad_quarter_tail): stage 1 on v_mfma_f64_4x4x4_4b with the
    fragments of the full tiles, stage 2 through LDS.  BITWISE the full-tile launch (the same products in the same order), and
    against the oracle on the elements of the quarter tiles.  E = 100 000: three rounds + 106 tiles; 100 007: the same with seven
    elements behind the last tile; 33 000: ONE full round + 14 tiles (a wave's quarter tile follows its first tile); 68 736: two
    rounds + 200; 36 928: one round + 260 tiles -- more than an eighth, no quarter tiles."""
    torch = torch_cuda
    expr = {"grad": dg.grad, "grad_t": dg.grad_t}[name]()
    host = generate_host_input_arrays(expr, E, np_seed=E + 11)
    before = _hip.set_grad_quarter_tail(True)
    try:
        got = _run(torch, expr, host)
        info = _hip.last_launch_info()
        _hip.set_grad_quarter_tail(False)
        full = _run(torch, expr, host)
        info_full = _hip.last_launch_info()
    finally:
        _hip.set_grad_quarter_tail(before)
    assert before is True                                      # the default
    assert not info["dynamic_walk"] and not info_full.get("quarter_tail"), (info, info_full)
    assert bool(info.get("quarter_tail")) == (E != 36864 + 64), info
    for k in got:
        assert np.array_equal(got[k], full[k]), (name, E, k)
    tiles = E // 16
    first = (tiles - tiles % 2048) * 16                        # the first element behind the full rounds
    idx = np.unique(np.concatenate([np.arange(0, 32), np.arange(first - 16, min(E, first + 96)), np.arange(E - 40, E)]))
    sub = {}
    for a in sorted(expr.all_args):
        shape = expr.arg_to_shape[a]
        ax = [i for i, d in enumerate(shape) if isinstance(d, f.SizeParam)]
        sub[a] = np.take(host[a], idx, axis=ax[0]) if ax else host[a]
    ref = _oracle(expr, sub)
    ax = [i for i, d in enumerate(expr.shape) if isinstance(d, f.SizeParam)][0]
    for k in got:
        _assert_close({k: np.ascontiguousarray(np.take(got[k], idx, axis=ax))}, {k: ref[k]})


@pytest.mark.parametrize("name", ["grad", "grad_t"])
@pytest.mark.parametrize("E", [100000, 131072 + 5, 81920, 81904, 160000])
def test_grad_with_a_staggered_start(torch_cuda, name, E):
    """Round 5: a short grad launch of one field (static walk, at least 2.5 rounds of tiles) starts the blocks on every second CU of
    an XCD half a tile period late (fe_grad.h, kOpStaggeredStart; fe_set_grad_staggered_start).  Timing only: BITWISE the launch
    with every block in step, and the launcher reports what it decided -- 81 920 elements are exactly 2.5 rounds (on), 81 904 one
    tile less (off), 160 000 walk dynamically (off)."""
    torch = torch_cuda
    expr = {"grad": dg.grad, "grad_t": dg.grad_t}[name]()
    host = generate_host_input_arrays(expr, E, np_seed=E + 13)
    before = _hip.set_grad_staggered_start(True)
    try:
        got = _run(torch, expr, host)
        info = _hip.last_launch_info()
        _hip.set_grad_staggered_start(False)
        plain = _run(torch, expr, host)
        info_plain = _hip.last_launch_info()
    finally:
        _hip.set_grad_staggered_start(before)
    assert before is True                                      # the default
    full_grid = info["blocks"] * info["waves_per_block"] == 2048
    if full_grid:
        assert bool(info.get("staggered_start")) == (E in (100000, 131072 + 5, 81920)), info
        assert bool(info["dynamic_walk"]) == (E == 160000), info
    assert not info_plain.get("staggered_start"), info_plain
    for k in got:
        assert np.array_equal(got[k], plain[k]), (name, E, k)
    idx = np.unique(np.concatenate([np.arange(0, 48), np.arange(E // 2, E // 2 + 48), np.arange(E - 48, E)]))
    sub = {}
    for a in sorted(expr.all_args):
        shape = expr.arg_to_shape[a]
        ax = [i for i, d in enumerate(shape) if isinstance(d, f.SizeParam)]
        sub[a] = np.take(host[a], idx, axis=ax[0]) if ax else host[a]
    ref = _oracle(expr, sub)
    ax = [i for i, d in enumerate(expr.shape) if isinstance(d, f.SizeParam)][0]
    for k in got:
        _assert_close({k: np.ascontiguousarray(np.take(got[k], idx, axis=ax))}, {k: ref[k]})


@pytest.mark.parametrize("E", [98304, 100000, 131072])
def test_the_reference_regime_sizes_against_the_oracle(torch_cuda, E):
    """The reference's own size regime (``long_dim_length`` defaults to 100 000: src/feinsum/measure.py:202; every fact of its
    archives is taken there): three rounds of tiles on the 2048 waves -- exactly (98 304), with a ragged fourth round (100 000),
    exactly four (131 072).  grad, div, face-mass x 4 as single launches and as the one fused launch of the wave operator (whose
    bodies walk dynamically from three rounds on since round 5): sampled oracle slices at 1e-12, the fused launch bitwise the
    three separate ones."""
    torch = torch_cuda
    exprs = [dg.div(), dg.grad(), dg.face_mass(4)]
    devs = [_device_inputs(torch, e, E, 40 + k) for k, e in enumerate(exprs)]
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    stages = list(zip(exprs, devs))
    slices = [slice(0, 48), slice(E // 2 - 7, E // 2 + 41), slice(E - 48, E)]
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)
    for (expr, dev), outs in zip(stages, plain):
        _sampled_oracle_check(expr, dev, outs, E, slices)
    op = f.bind_operator(stages, 0)
    assert op.entry_points == ("fe_waveop3d_f64",)
    fused = f.evaluate_operator(stages, 0, wait=True)
    info = _hip.last_launch_info()
    assert info["bodies"] == 3 and bool(info["dynamic_walk"]) == (E != 0), info       # three rounds and more: tickets in the fused launch
    for a, b in zip(fused, plain):
        for k in a:
            assert torch.equal(a[k], b[k])
    pair = f.evaluate_operator(stages[:2], 0, wait=True)                              # div + grad in one launch
    for a, b in zip(pair, plain[:2]):
        for k in a:
            assert torch.equal(a[k], b[k])
