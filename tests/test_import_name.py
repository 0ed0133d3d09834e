"""`import feinsum as f`: the reference's own spelling of its tests runs on this backend
(test/test_codegen.py:96-120, test/test_measure.py:55-81 with cq -> a device ordinal)."""

import numpy as np
import pytest


def test_builder_api_under_the_reference_import_name():
    import feinsum as f
    from feinsum.einsum import BatchedEinsum, SizeParam
    from feinsum.measure import _get_giga_ops_from_einsum

    Ndim, Ndof = 3, 35
    expr = f.einsum("xre,rij,ej->xei", f.array("J", (Ndim, Ndim, "Nel")), f.array("R", (Ndim, Ndof, Ndof)),
                    f.array("u", ("Nel", Ndof)))
    assert isinstance(expr, BatchedEinsum) and isinstance(expr.shape[1], SizeParam)
    assert f.get_opt_einsum_contraction_schedule(expr).nsteps == 2
    assert _get_giga_ops_from_einsum(expr, 100_000)[np.dtype("float64")] == pytest.approx(0.798)
    import feinsum.sql_utils as sql
    assert hasattr(sql, "record_facts") and f.array is not None
    with pytest.raises(f.NoDevicePeaksInfoError):
        f.get_roofline_flop_rate(expr, "no such device")


@pytest.mark.gpu
def test_reference_style_measure_calls():
    import feinsum as f

    Ndim, Ndof = 3, 35
    expr = f.einsum("xre,rij,ej->xei", f.array("J", (Ndim, Ndim, "Nel")), f.array("R", (Ndim, Ndof, Ndof)),
                    f.array("u", ("Nel", Ndof)))
    identity = lambda t_unit, insn_match=None, kernel_name=None: t_unit   # noqa: E731  (the reference's transform slot)
    assert f.timeit(expr, cq=0, transform=identity, long_dim_length=300) > 0
    table = f.stringify_comparison_vs_roofline(expr, cq=0, transform=identity, long_dim_length=500)
    assert "Measured GOps/s" in table and "N/A" not in table
