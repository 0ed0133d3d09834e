"""Known-answer pins of the hot path: op counts, schedules,
footprint bytes and the roofline formula."""

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd import measure
from feinsum_amd.contraction_schedule import EinsumOperand, IntermediateResult

import dg


def test_grad_flops_known_answer():
    # test/test_loopy_utils.py:267-271: 33075 trivial, 7980 hoisted per element
    g = dg.grad()
    assert f.count_ops(g, f.get_trivial_contraction_schedule(g)) == 33075
    assert f.count_ops(g) == 7980
    assert f.count_ops(g, long_dim_length=1000) == 7980 * 1000


def test_archive_giga_op_info():
    # data/transform_archive_v5.sqlite giga_op_info at E = 1e5: 0.798 / 0.798 / 1.704
    E = 100_000
    assert f.count_ops(dg.grad(), long_dim_length=E) * 1e-9 == pytest.approx(0.798)
    assert f.count_ops(dg.div(), long_dim_length=E) * 1e-9 == pytest.approx(0.798)
    assert f.count_ops(dg.face_mass(), long_dim_length=E) * 1e-9 == pytest.approx(1.704)
    assert f.count_ops(dg.face_mass()) == 17040
    assert f.count_ops(dg.face_mass_ifj_fe()) == 17040


def test_optimal_schedule_is_two_steps():
    # test/test_codegen.py:134-137: 1 instruction trivial, 2 with the opt_einsum schedule
    g = dg.grad()
    assert f.get_trivial_contraction_schedule(g).nsteps == 1
    s = f.get_opt_einsum_contraction_schedule(g)
    assert s.nsteps == 2 and s.result_names == ("_fe_tmp", "_fe_out")
    # D.u first, then the Jacobian combine (ej,rij->rie ; rie,xre->xei)
    first = {a.ioperand for a in s.arguments[0]}
    assert first == {1, 2}
    assert any(isinstance(a, IntermediateResult) for a in s.arguments[1])
    assert EinsumOperand(0) in s.arguments[1]
    d = f.get_opt_einsum_contraction_schedule(dg.div())
    assert {a.ioperand for a in d.arguments[0]} == {0, 2}      # J.u first
    m = f.get_opt_einsum_contraction_schedule(dg.face_mass())
    assert {a.ioperand for a in m.arguments[0]} == {0, 2}      # J.v first


def test_schedules_evaluate_to_the_same_values():
    # the 2-step schedule is value-preserving (test/test_codegen.py:137 auto_test_vs_ref, E=5)
    rng = np.random.default_rng(1)
    J, D, u = rng.random((3, 3, 5)), rng.random((3, 35, 35)), rng.random((5, 35))
    s = f.get_opt_einsum_contraction_schedule(dg.grad())
    env = {}
    ops = [J, D, u]
    for subs, name, args in zip(s.subscripts, s.result_names, s.arguments):
        vals = [ops[a.ioperand] if isinstance(a, EinsumOperand) else env[a.name] for a in args]
        env[name] = np.einsum(subs, *vals)
    np.testing.assert_allclose(env["_fe_out"], np.einsum("xre,rij,ej->xei", J, D, u), rtol=1e-13)


def test_footprint_and_roofline():
    g = dg.grad()
    # measure.py:334-354: every arg once; grad = 8*(149 E + 3675) bytes
    E = 100_000
    assert measure._get_footprint_gbytes(g, E) == pytest.approx(8 * (149 * E + 3675) * 1e-9)
    # TITAN V roofline 0.11920 GB / 652.8 GB/s -> 4370 GFLOP/s
    r = f.get_roofline_flop_rate(g, "NVIDIA TITAN V", E)
    assert r[np.dtype("float64")] == pytest.approx(4370, rel=2e-3)
    # MI355X: HBM-bound, min(78.6 TF, 6.69 flop/B * 8 TB/s) = 53.6 TFLOP/s
    r = f.get_roofline_flop_rate(g, "AMD Instinct MI355X", 1_000_000)
    assert r[np.dtype("float64")] == pytest.approx(53_556, rel=1e-3)
    fm = f.get_roofline_flop_rate(dg.face_mass(), "AMD Instinct MI355X", 1_000_000)
    assert fm[np.dtype("float64")] == pytest.approx(17040 / 3072 * 8000, rel=1e-3)
    with pytest.raises(f.NoDevicePeaksInfoError):
        f.get_roofline_flop_rate(g, "no such device")


def test_roofline_table_format():
    s = measure._strify_measured_vs_roofline({np.dtype("float64"): 1234.56},
                                             {np.dtype("float64"): "N/A"})
    assert "Measured GOps/s" in s and "1234.6" in s and "N/A" in s and "float64" in s
