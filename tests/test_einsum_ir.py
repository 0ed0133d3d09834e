"""Builders + IR: behaviour of feinsum.einsum / feinsum.make_einsum
(reference: src/feinsum/einsum.py:159-387, src/feinsum/make_einsum.py:55-156)."""

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd.einsum import SizeParam

import dg


def test_array_builder():
    a = f.array("u", ("E", 35))
    assert a.shape == (SizeParam("E"), 35) and a.dtype == np.float64 and a.ndim == 2
    assert f.array("x", 4, "float32").shape == (4,)
    assert f.array("x", 4, "float32").dtype == np.float32
    assert a.copy(name="w").name == "w" and a.copy(name="w").shape == a.shape
    with pytest.raises(ValueError):
        f.array("x", (-1, 3))
    with pytest.raises(ValueError):
        f.array("x", (2.5,))


def test_grad_properties():
    e = dg.grad()
    assert (e.b, e.n, e.ndim) == (1, 3, 3)
    assert e.shape == (3, SizeParam("E"), 35)
    assert e.out_idx_set == ("x", "e", "i")
    assert e.in_idx_sets == (("x", "r", "e"), ("r", "i", "j"), ("e", "j"))
    assert e.sum_indices == ("r", "j")
    assert e.get_subscripts() == "xre,rij,ej -> xei"
    assert e.all_args == frozenset({"J", "R", "u"})
    assert e.all_indices == frozenset("xreij")
    assert e.all_size_params == frozenset({SizeParam("E")})
    assert dict(e.index_to_dim_length)["j"] == 35
    assert dict(e.arg_to_shape)["J"] == (3, 3, SizeParam("E"))
    assert e.output_names == ("_fe_out",)
    acc = e.index_to_access_descr
    assert acc["x"] == f.FreeAxis(0) and acc["i"] == f.FreeAxis(2)
    assert acc["r"] == f.SummationAxis(0) and acc["j"] == f.SummationAxis(1)
    assert "Σ_{r, j} J[x, r, e]×R[r, i, j]×u[e, j]" in str(e)
    assert hash(e) == hash(dg.grad()) and e == dg.grad()


def test_batched_properties():
    e = dg.face_mass()
    assert (e.b, e.n) == (4, 3)
    assert e.output_names == ("_fe_out", "_fe_out_0", "_fe_out_1", "_fe_out_2")
    assert e.all_args == frozenset({"J", "R", "v0", "v1", "v2", "v3"})
    assert e.shape == (SizeParam("E"), 35)
    c = e.copy(args=e.args[:2])
    assert c.b == 2 and c.in_idx_sets == e.in_idx_sets


def test_subscript_errors():
    A, x = f.array("A", (10, 4)), f.array("x", 4)
    with pytest.raises(ValueError, match="Missing ->"):
        f.einsum("ij,j", A, x)
    with pytest.raises(NotImplementedError):
        f.einsum("...j,j->...", A, x)
    with pytest.raises(ValueError, match="Cannot parse"):
        f.einsum("i1,j->i", A, x)
    with pytest.raises(ValueError, match="more than once"):
        f.einsum("ij,j->ii", A, x)
    # AssertionError -> TypeError (make_einsum.py:143-148)
    with pytest.raises(TypeError, match="Dimensionality"):
        f.einsum("ijk,j->i", A, x)
    with pytest.raises(TypeError, match="#operands"):
        f.einsum("ij->i", A, x)
    with pytest.raises(TypeError, match="not present in the input"):
        f.einsum("ij,j->k", A, x)
    with pytest.raises(TypeError, match="invalid input index"):
        f.einsum("Ij,j->j", A, x)
    with pytest.raises(TypeError, match="Shape mismatch"):
        f.einsum("ij,j->i", A, f.array("x", 5))
    with pytest.raises(TypeError, match="Inconsistent shapes"):
        f.batched_einsum("ij,j->i", [[A, x], [f.array("A", (10, 4)), f.array("x", 4)],
                                     [f.array("A", (9, 4)), x]])
    with pytest.raises(TypeError, match="Inconsistent dtypes"):
        f.batched_einsum("ij,j->i", [[A, x], [A, f.array("x", 4, "float32")]])
    with pytest.raises(TypeError, match="different names"):
        f.einsum("ij,j->i", f.array("i", (10, 4)), x)
    with pytest.raises(TypeError):
        f.EinsumAxisAccess()


def test_whitespace_and_parametric_matvec():
    # test/test_measure.py:33-52
    A = f.array("A", ("I", 4), "float32")
    e = f.batched_einsum("ij, j -> i", [[A, f.array("x", 4, "float32")],
                                        [A, f.array("y", 4, "float32")]])
    assert e.b == 2 and e.shape == (SizeParam("I"),)
    assert e.get_subscripts() == "ij,j -> i"


def test_api_parity_helpers():
    from feinsum_amd.typing import TransformT

    assert f.FakeCLDevice("AMD Instinct MI355X").name == "AMD Instinct MI355X"
    ident: TransformT = lambda t_unit, insn_match=None, kernel_name=None: t_unit  # noqa: E731
    assert ident(1) == 1
    r = f.get_roofline_flop_rate(dg.grad(), f.FakeCLDevice("AMD Instinct MI355X").name, 1_000_000)
    assert r[np.dtype("float64")] > 5e4
