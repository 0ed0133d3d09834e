"""The oracle against the committed golden vectors, and its three restatements
against each other (numpy ground truth / extended-precision sum of products /
plain-C loop nests)."""

import numpy as np
import pytest

from oracle import c_oracle, np_oracle

import dg

TOL = 1e-12   # north_star: 1e-12 relative


def _load(golden_dir, name, E):
    z = np.load(golden_dir / f"{name}_E{E}.npz")
    return {k[3:]: z[k] for k in z.files if k.startswith("in_")}, \
           {k[4:]: z[k] for k in z.files if k.startswith("out_")}, str(z["subscripts"])


@pytest.mark.parametrize("name", sorted(dg.GOLDEN_CASES))
@pytest.mark.parametrize("E", [1, 7, 37])
def test_oracle_reproduces_golden(golden_dir, name, E):
    expr = dg.GOLDEN_CASES[name]()
    ins, outs, subs = _load(golden_dir, name, E)
    assert subs == expr.get_subscripts()
    assert set(ins) == set(expr.all_args) and set(outs) == set(expr.output_names)
    for out_name, row in zip(expr.output_names, expr.args):
        ops = [ins[a.name] for a in row]
        got = np_oracle.reference_outputs(subs, [ops])[0]
        assert got.shape == outs[out_name].shape
        assert np_oracle.max_rel_err(got, outs[out_name]) <= 1e-15
        assert np_oracle.max_rel_err(np_oracle.naive_longdouble(subs, ops), outs[out_name]) <= TOL


def test_golden_inputs_are_the_documented_stream(golden_dir):
    # default_rng(0), sorted-argument-name order, uniform[0,1)
    from feinsum_amd.measure import generate_host_input_arrays
    for name in ("grad_p4", "facemass_p4_ifj_fe"):
        expr = dg.GOLDEN_CASES[name]()
        ins, _, _ = _load(golden_dir, name, 7)
        regen = generate_host_input_arrays(expr, 7)
        assert list(regen) == sorted(regen)
        for k in ins:
            np.testing.assert_array_equal(ins[k], regen[k])
            assert ins[k].min() >= 0.0 and ins[k].max() < 1.0


def test_pure_python_loop_nest_tiny(golden_dir):
    ins, outs, subs = _load(golden_dir, "grad_p2", 1)
    got = np_oracle.loop_reference(subs, [ins["J"], ins["R"], ins["u"]])
    assert np_oracle.max_rel_err(got, outs["_fe_out"]) <= TOL


@pytest.mark.parametrize("kind", ["trivial", "hoisted"])
def test_c_loop_nests_match_golden(golden_dir, kind):
    c_oracle.build()
    ins, outs, _ = _load(golden_dir, "grad_p4", 37)
    assert np_oracle.max_rel_err(c_oracle.grad3d(ins["J"], ins["R"], ins["u"], kind), outs["_fe_out"]) <= TOL
    ins, outs, _ = _load(golden_dir, "div_p4", 37)
    assert np_oracle.max_rel_err(c_oracle.div3d(ins["J"], ins["R"], ins["u"], kind), outs["_fe_out"]) <= TOL
    ins, outs, _ = _load(golden_dir, "facemass_p4_ef_fij", 37)
    for k, out_name in enumerate(("_fe_out", "_fe_out_0", "_fe_out_1", "_fe_out_2")):
        got = c_oracle.facemass(ins["J"], ins["R"], ins[f"v{k}"], kind)
        assert np_oracle.max_rel_err(got, outs[out_name]) <= TOL
    ins, outs, _ = _load(golden_dir, "facemass_p4_ifj_fe", 37)
    got = c_oracle.facemass(ins["J"], ins["L"], ins["v2"], kind, jfe=True, rifj=True)
    assert np_oracle.max_rel_err(got, outs["_fe_out_1"]) <= TOL


def test_c_oracle_empty_and_single():
    c_oracle.build()
    rng = np.random.default_rng(3)
    D = rng.random((3, 35, 35))
    assert c_oracle.grad3d(np.empty((3, 3, 0)), D, np.empty((0, 35))).shape == (3, 0, 35)
    J, u = rng.random((3, 3, 1)), rng.random((1, 35))
    ref = np_oracle.reference_outputs("xre,rij,ej->xei", [[J, D, u]])[0]
    assert np_oracle.max_rel_err(c_oracle.grad3d(J, D, u), ref) <= TOL
