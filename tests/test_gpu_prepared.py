"""Prepared operators (fe_prepare_operator): a launch that takes its MFMA fragments from the
prepared copy is bitwise the launch that rebuilds them from the plain array, for every family,
order, layout and field count that has a prepared form; and the launchers refuse buffers that do
not fit the call."""

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd import _hip, measure
from feinsum_amd.measure import generate_host_input_arrays

import dg

pytestmark = pytest.mark.gpu
TOL = 1e-12
ORDERS = [(4, 3), (10, 6), (20, 10), (35, 15)]


def _dev(torch, host):
    return {k: torch.from_numpy(v).cuda() for k, v in host.items()}


def _bound_outputs(torch, expr, dev, prepare, transform=None):
    q, bound, outs = measure._bind(expr, 0, dev, None, transform, prepare=prepare)
    bound.launch(q.stream_ptr)
    q.finish()
    return bound, [o.clone() for o in outs]


@pytest.mark.parametrize("Np,Nfp", ORDERS)
@pytest.mark.parametrize("E", [15, 16, 17, 100, 1003, 4099, 33000])
def test_prepared_launches_are_bitwise_the_plain_ones(Np, Nfp, E):
    import torch

    from oracle import np_oracle

    exprs = [dg.grad(Np), dg.grad_t(Np), dg.batched_grad(3, Np), dg.div(Np), dg.div_t(Np), dg.batched_div(2, Np),
             dg.face_mass(4, Np=Np, Nfp=Nfp), dg.face_mass_ifj_fe(3, Np=Np, Nfp=Nfp), dg.face_mass_jfi_fe(2, Np=Np, Nfp=Nfp),
             dg.face_mass_fji(5, Np=Np, Nfp=Nfp)]
    for expr in exprs:
        host = generate_host_input_arrays(expr, E, np_seed=E + Np)
        dev = _dev(torch, host)
        plain_bound, plain = _bound_outputs(torch, expr, dev, prepare=False)
        prep_bound, prep = _bound_outputs(torch, expr, dev, prepare=True)
        assert all(p.prepared is None for p in plain_bound.groups)
        assert all(p.prepared for p in prep_bound.groups), expr.get_subscripts()
        for a, b in zip(prep, plain):
            assert torch.equal(a, b), expr.get_subscripts()
        for name, row, got in zip(expr.output_names, expr.args, prep):
            ref = np_oracle.reference_outputs(expr.get_subscripts(), [[host[x.name] for x in row]])[0]
            assert np_oracle.max_rel_err(got.cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("Np,Nfp", ORDERS)
@pytest.mark.parametrize("E", [7, 1003, 40_009])
def test_prepared_fused_launches(Np, Nfp, E):
    """graddiv / waveop with prepared operators: bitwise the separate unprepared launches."""
    import torch

    exprs = [dg.div(Np), dg.grad(Np), dg.face_mass(4, Np=Np, Nfp=Nfp)]
    hosts = [generate_host_input_arrays(e, E, np_seed=k) for k, e in enumerate(exprs)]
    devs = [_dev(torch, h) for h in hosts]
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    stages = list(zip(exprs, devs))
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)          # evaluate_operator never prepares
    for sub in (stages[:2], stages):
        op = f.bind_operator(sub, 0, prepare=True)
        assert len(op.entry_points) == 1
        assert all(b._prepared for b in op._stages)
        op.launch()
        op.queue.finish()
        for got, ref in zip(op.outputs, plain):
            for k in got:
                assert torch.equal(got[k], ref[k])
        op2 = f.bind_operator(sub, 0)                                       # off by default
        assert not any(b._prepared for b in op2._stages)


def test_refresh_after_changing_the_operator_in_place():
    import torch

    from oracle import np_oracle

    E = 2000                                   # (whole tiles: the elements behind the last tile read the plain array)
    expr = dg.grad()
    host = generate_host_input_arrays(expr, E)
    dev = _dev(torch, host)
    op = f.bind_operator([(expr, dev)], 0, prepare=True)
    op.launch(); op.queue.finish()
    first = op.outputs[0]["_fe_out"].clone()
    dev["R"].mul_(2.0)                         # the operator changes in place ...
    op.launch(); op.queue.finish()
    assert torch.equal(op.outputs[0]["_fe_out"], first)     # ... the prepared snapshot does not (documented)
    op.refresh_operators()
    op.launch(); op.queue.finish()
    assert torch.equal(op.outputs[0]["_fe_out"], first * 2.0)
    ref = np.einsum("xre,rij,ej->xei", host["J"], 2.0 * host["R"], host["u"], optimize="optimal")
    assert np_oracle.max_rel_err(op.outputs[0]["_fe_out"].cpu().numpy(), ref) <= TOL


def test_buffers_that_do_not_fit_the_call_are_refused():
    import torch

    from feinsum_amd.diagnostics import InvalidParameterError

    E, Np = 64, 35
    host = generate_host_input_arrays(dg.grad(), E)
    J, D, u = (torch.from_numpy(host[k]).cuda() for k in ("J", "R", "u"))
    out = torch.empty((3, E, Np), dtype=torch.float64, device="cuda")
    lib = _hip.load_library()
    buf = torch.empty(_hip.PREPARED_OPERATOR_BYTES, dtype=torch.uint8, device="cuda")
    up, op_ = _hip._ptr_array([u.data_ptr()]), _hip._ptr_array([out.data_ptr()])

    def grad(prepared, np_=Np, flags=0):
        _hip.check(lib.fe_grad3d_prepared_f64(J.data_ptr(), D.data_ptr(), prepared, up, op_, E, np_, 1, flags, 0, 0))

    # (torch may hand out an address an earlier test prepared for something else: refused either way)
    with pytest.raises(InvalidParameterError, match="not written by fe_prepare_operator|another call"):
        grad(buf.data_ptr())
    _hip.prepare_operator(1, D.data_ptr(), Np, 0, 0, 0, buf.data_ptr())
    grad(buf.data_ptr())
    torch.cuda.synchronize()
    with pytest.raises(InvalidParameterError, match="another call"):
        grad(buf.data_ptr(), flags=1)                        # prepared for the untransposed layout
    # a snapshot of ANOTHER operator array of the same shape and flags is refused too (it would compute with that one)
    D_other = D.clone()
    with pytest.raises(InvalidParameterError, match="snapshot of the operator array"):
        _hip.check(lib.fe_grad3d_prepared_f64(J.data_ptr(), D_other.data_ptr(), buf.data_ptr(), up, op_, E, Np, 1, 0, 0, 0))
    with pytest.raises(InvalidParameterError, match="another call"):      # a D buffer is not an R buffer
        v = torch.zeros((4, E, 15), dtype=torch.float64, device="cuda")
        o = torch.zeros((E, Np), dtype=torch.float64, device="cuda")
        _hip.check(lib.fe_facemass_prepared_f64(J.data_ptr(), D.data_ptr(), buf.data_ptr(),
                                                _hip._ptr_array([v.data_ptr()] * 2), _hip._ptr_array([o.data_ptr()] * 2),
                                                E, Np, 4, 15, 2, 0, 0, 0))
    with pytest.raises(NotImplementedError):
        _hip.prepare_operator(1, D.data_ptr(), 56, 0, 0, 0, buf.data_ptr())       # p = 5: no prepared form
    # variants without a prepared form ignore the buffer
    _hip.check(lib.fe_grad3d_prepared_f64(J.data_ptr(), D.data_ptr(), buf.data_ptr(), up, op_, E, Np, 1, 0, 1, 0))
    torch.cuda.synchronize()


def test_timeit_options_prepared_and_placement():
    expr = dg.grad()
    t_plain = measure.timeit_details(expr, cq=0, long_dim_length=100_000, min_secs=0.2)
    t_prep = measure.timeit_details(expr, cq=0, long_dim_length=100_000, min_secs=0.2, transform={"prepared": True})
    for t in (t_plain, t_prep):
        assert 0 < t.seconds_device < 1e-3
    # the result says how its arrays were placed (a recorded fact must be reproducible by a caller)
    assert t_plain.placement["mode"] == "split" and t_prep.placement["mode"] == "split"
    t_sep = measure.timeit_details(expr, cq=0, long_dim_length=100_000, min_secs=0.2, transform={"placement": "separate"})
    assert t_sep.placement["mode"] == "separate" and 0 < t_sep.seconds_device < 1e-3
    with pytest.raises(f.InvalidParameterError):        # round 2's arena scan is gone
        measure.timeit_details(expr, cq=0, long_dim_length=100_000, min_secs=0.2, transform={"placement": "tuned"})


def test_release_drops_the_record_of_a_prepared_buffer():
    """fe_release_prepared: afterwards the address is no prepared operator any more (a later unrelated allocation at the
    same address must not be taken for one: ADVICE r02); bound launches release their buffers when they go away."""
    import gc

    import torch

    from feinsum_amd.diagnostics import InvalidParameterError

    E, Np = 64, 35
    host = generate_host_input_arrays(dg.grad(), E)
    J, D, u = (torch.from_numpy(host[k]).cuda() for k in ("J", "R", "u"))
    out = torch.empty((3, E, Np), dtype=torch.float64, device="cuda")
    lib = _hip.load_library()
    buf = torch.empty(_hip.PREPARED_OPERATOR_BYTES, dtype=torch.uint8, device="cuda")
    up, op_ = _hip._ptr_array([u.data_ptr()]), _hip._ptr_array([out.data_ptr()])
    _hip.prepare_operator(1, D.data_ptr(), Np, 0, 0, 0, buf.data_ptr())
    _hip.check(lib.fe_grad3d_prepared_f64(J.data_ptr(), D.data_ptr(), buf.data_ptr(), up, op_, E, Np, 1, 0, 0, 0))
    torch.cuda.synchronize()
    _hip.release_prepared(buf.data_ptr())
    with pytest.raises(InvalidParameterError, match="not written by fe_prepare_operator"):
        _hip.check(lib.fe_grad3d_prepared_f64(J.data_ptr(), D.data_ptr(), buf.data_ptr(), up, op_, E, Np, 1, 0, 0, 0))
    with pytest.raises(InvalidParameterError):
        _hip.release_prepared(buf.data_ptr())                # no record left
    # a bound launch with prepared operators releases its records with itself
    dev = _dev(torch, host)
    op = f.bind_operator([(dg.grad(), dev)], 0, prepare=True)
    ptrs = [b.data_ptr() for st in op._stages for b in st._prepared.values()]
    assert ptrs
    del op
    gc.collect()
    for ptr in ptrs:
        with pytest.raises(InvalidParameterError):
            _hip.release_prepared(ptr)
