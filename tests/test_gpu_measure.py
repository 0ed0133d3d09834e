"""feinsum.measure API on the device: the calls the reference's own tests make
(test/test_codegen.py:34-120, test/test_measure.py:33-81), the error behaviour,
and the fused grad+div entry point of the C ABI."""

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd import _hip, measure
from feinsum_amd.measure import generate_host_input_arrays

import dg

pytestmark = pytest.mark.gpu
IDENTITY = lambda t_unit, insn_match, kernel_name: t_unit  # noqa: E731  (the reference's identity transform)


@pytest.fixture(autouse=True)
def _fast_protocol(monkeypatch):
    # keep the suite short: the protocol is the reference's, only the 2 s floor is lowered
    monkeypatch.setattr(measure, "N_MIN_SIM_SECS", 0.05)


def _timeit(expr, **kw):
    return measure.timeit_details(expr, min_secs=0.05, **kw)


def test_timeit_like_reference_tests():
    import torch

    cq = f.DeviceQueue(0)
    assert "MI355X" in cq.device.name or cq.device.name
    for expr in (dg.batched_div_components(), dg.face_mass(), dg.grad()):
        r = _timeit(expr, transform=IDENTITY, cq=cq, long_dim_length=300)
        assert r.rounds >= 10 and r.rounds % 5 == 0
        assert 0 < r.seconds_device <= r.seconds_wall * 1.5
    t = f.timeit(dg.grad(), transform=IDENTITY, cq=0, long_dim_length=3000)
    assert 0 < t < 1e-2
    torch.cuda.synchronize()


def test_matvec_float32_fixed_and_parametric():
    for A in (f.array("A", (10, 4), "float32"), f.array("A", ("I", 4), "float32")):
        expr = f.batched_einsum("ij, j -> i", [[A, f.array("x", 4, "float32")],
                                               [A, f.array("y", 4, "float32")]])
        r = _timeit(expr, transform=IDENTITY, cq=0, long_dim_length=1000)
        assert r.seconds_device > 0


def test_pprint_roofline_comparison():
    s = f.stringify_comparison_vs_roofline(dg.grad(), cq=0, transform=IDENTITY, long_dim_length=500)
    assert "Measured GOps/s" in s and "Roofline GOps/s" in s and "float64" in s
    assert "N/A" not in s          # the MI355X row exists in the device table
    rates = f.measure_giga_op_rate(dg.div(), cq=0, transform=None, long_dim_length=20000)
    assert list(rates) == [np.dtype("float64")] and rates[np.dtype("float64")] > 0


def test_validation_catches_a_wrong_kernel(monkeypatch):
    # a "transform" that yields wrong values must raise TransformValidationError
    real = measure.evaluate

    def broken(einsum, cq, arg_dict, **kw):
        outs = real(einsum, cq, arg_dict, **kw)
        for o in outs.values():
            o[..., 0] *= 1.0 + 1e-8
        return outs

    monkeypatch.setattr(measure, "evaluate", broken)
    with pytest.raises(f.TransformValidationError):
        f.validate_batched_einsum_transform(dg.grad(), 0, None)
    monkeypatch.undo()
    f.validate_batched_einsum_transform(dg.grad(), 0, None)   # and passes unbroken


def test_evaluate_argument_errors():
    import torch

    expr = dg.grad()
    host = generate_host_input_arrays(expr, 64)
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    with pytest.raises(f.InvalidParameterError, match="missing"):
        f.evaluate(expr, 0, {"J": dev["J"]})
    with pytest.raises(f.InvalidParameterError, match="shape"):
        f.evaluate(expr, 0, {**dev, "R": dev["R"][:, :34].contiguous()})
    with pytest.raises(f.InvalidParameterError, match="dtype"):
        f.evaluate(expr, 0, {**dev, "u": dev["u"].float()})
    with pytest.raises(f.InvalidParameterError, match="contiguous"):
        f.evaluate(expr, 0, {**dev, "u": torch.empty((35, 64), dtype=torch.float64, device="cuda").t()})
    with pytest.raises(f.InvalidParameterError, match="lives on"):
        f.evaluate(expr, 0, {**dev, "u": dev["u"].cpu()})
    with pytest.raises(f.InvalidParameterError, match="inconsistent"):
        f.evaluate(expr, 0, {**dev, "u": torch.zeros((65, 35), dtype=torch.float64, device="cuda")})


@pytest.mark.parametrize("Np", [4, 10, 20, 35, 56])
@pytest.mark.parametrize("E", [1, 15, 80, 4099, 20011])
def test_fused_graddiv_single_launch(Np, E):
    # div then grad in one persistent launch (BASELINE config 3); tails and orders without an
    # MFMA geometry go through the generic kernels
    import torch

    from oracle import np_oracle

    rng = np.random.default_rng(Np + E)
    J, D = rng.random((3, 3, E)), rng.random((3, Np, Np))
    u, v = rng.random((E, Np)), rng.random((3, E, Np))
    dJ, dD, du, dv = (torch.from_numpy(a).cuda() for a in (J, D, u, v))
    og = torch.full((3, E, Np), float("nan"), dtype=torch.float64, device="cuda")
    od = torch.full((E, Np), float("nan"), dtype=torch.float64, device="cuda")
    _hip.graddiv3d(dJ.data_ptr(), dD.data_ptr(), du.data_ptr(), dv.data_ptr(), og.data_ptr(),
                   od.data_ptr(), E, Np, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np_oracle.max_rel_err(og.cpu().numpy(), np.einsum("xre,rij,ej->xei", J, D, u, optimize="optimal")) <= 1e-12
    assert np_oracle.max_rel_err(od.cpu().numpy(), np.einsum("xre,rij,xej->ei", J, D, v, optimize="optimal")) <= 1e-12
    # bitwise the separate launches
    og2, od2 = torch.empty_like(og), torch.empty_like(od)
    _hip.grad3d(dJ.data_ptr(), dD.data_ptr(), du.data_ptr(), og2.data_ptr(), E, Np)
    _hip.div3d(dJ.data_ptr(), dD.data_ptr(), dv.data_ptr(), od2.data_ptr(), E, Np)
    assert torch.equal(og, og2) and torch.equal(od, od2)


def test_fused_graddiv_and_time_launches():
    import torch

    from oracle import np_oracle

    E = 5000
    rng = np.random.default_rng(9)
    J, D = rng.random((3, 3, E)), rng.random((3, 35, 35))
    u, v = rng.random((E, 35)), rng.random((3, E, 35))
    dJ, dD, du, dv = (torch.from_numpy(a).cuda() for a in (J, D, u, v))
    og = torch.empty((3, E, 35), dtype=torch.float64, device="cuda")
    od = torch.empty((E, 35), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _hip.graddiv3d(dJ.data_ptr(), dD.data_ptr(), du.data_ptr(), dv.data_ptr(), og.data_ptr(),
                   od.data_ptr(), E, 35, stream=s)
    torch.cuda.synchronize()
    assert np_oracle.max_rel_err(og.cpu().numpy(), np.einsum("xre,rij,ej->xei", J, D, u, optimize="optimal")) <= 1e-12
    assert np_oracle.max_rel_err(od.cpu().numpy(), np.einsum("xre,rij,xej->ei", J, D, v, optimize="optimal")) <= 1e-12
    pack = _hip.ArgPack()
    pack.J, pack.D, pack.u, pack.out = dJ.data_ptr(), dD.data_ptr(), du.data_ptr(), og.data_ptr()
    pack.E, pack.Np, pack.variant = E, 35, 0
    ms = _hip.time_launches(1, pack, 5, s)
    assert 0 < ms < 100
    name, pf, pb = _hip.device_info(0)
    assert pf == pytest.approx(78643.2, rel=0.05) and pb == 8000.0 and name


@pytest.mark.parametrize("Np,Nfp", [(35, 15), (10, 6)])
@pytest.mark.parametrize("E", [7, 4099])
def test_wave_operator_single_launch(Np, Nfp, E):
    """div(v), grad(u) and the lift of four face fields described as three einsums
    (examples/wave_3d_p4_auto.py:16-63) run as ONE launch and match the oracle stage by stage."""
    import torch

    from oracle import np_oracle

    exprs = [dg.div(Np), dg.grad(Np), dg.face_mass_ifj_fe(4, Np=Np, Nfp=Nfp)]
    hosts = [generate_host_input_arrays(e, E, np_seed=k) for k, e in enumerate(exprs)]
    hosts[1]["J"], hosts[1]["R"] = hosts[0]["J"], hosts[0]["R"]        # one geometry, one operator
    devs = [{k: torch.from_numpy(v).cuda() for k, v in h.items()} for h in hosts]
    devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
    stages = list(zip(exprs, devs))
    op = f.bind_operator(stages, 0)
    assert op.entry_points == ("fe_waveop3d_f64",)
    outs = f.evaluate_operator(stages, 0, wait=True)
    plain = f.evaluate_operator(stages, 0, fuse=False, wait=True)
    for expr, host, out, ref_out in zip(exprs, hosts, outs, plain):
        for name, row in zip(expr.output_names, expr.args):
            ref = np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in row]])[0]
            assert np_oracle.max_rel_err(out[name].cpu().numpy(), ref) <= 1e-12
            assert torch.equal(out[name], ref_out[name])                # bitwise the separate launches

    # what cannot share a launch is enqueued stage by stage
    five = dg.face_mass(5, Np=Np, Nfp=Nfp)
    h5 = generate_host_input_arrays(five, E, np_seed=5)
    op = f.bind_operator(stages[:2] + [(five, {k: torch.from_numpy(v).cuda() for k, v in h5.items()})], 0)
    assert op.entry_points[0] == "fe_graddiv3d_f64" and set(op.entry_points[1:]) == {"fe_facemass"}
    own_j = dict(devs[1], J=devs[1]["J"].clone())
    op = f.bind_operator([stages[0], (exprs[1], own_j)], 0)
    assert op.entry_points == ("fe_div", "fe_grad")
    assert op.time_batch(2) > 0


@pytest.mark.parametrize("E", [4099, 40_009])
def test_laplacian_stages_keep_their_order(E):
    """div(grad(u)): the div stage reads what the grad stage writes, so the two must not share a
    persistent launch (its bodies run in turn without a grid barrier); results equal fuse=False
    and the oracle."""
    import torch

    from oracle import np_oracle

    grad, div = dg.grad(), dg.div()
    host = generate_host_input_arrays(grad, E)
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    grad_out = torch.zeros((3, E, 35), dtype=torch.float64, device="cuda")
    stages = [(grad, dev), (div, {"J": dev["J"], "R": dev["R"], "u": grad_out})]
    out_dicts = [{"_fe_out": grad_out}, None]
    op = f.bind_operator(stages, 0, out_dicts=out_dicts)
    assert op.entry_points == ("fe_grad", "fe_div")
    fused = f.evaluate_operator(stages, 0, out_dicts=out_dicts, wait=True)
    lap = fused[1]["_fe_out"].clone()
    grad_out.zero_()
    plain = f.evaluate_operator(stages, 0, out_dicts=out_dicts, fuse=False, wait=True)
    assert torch.equal(lap, plain[1]["_fe_out"])
    g_ref = np.einsum("xre,rij,ej->xei", host["J"], host["R"], host["u"], optimize="optimal")
    ref = np.einsum("xre,rij,xej->ei", host["J"], host["R"], g_ref, optimize="optimal")
    assert np_oracle.max_rel_err(lap.cpu().numpy(), ref) <= 1e-12
    # independent div and grad over the same geometry still share one launch
    other = torch.rand((3, E, 35), dtype=torch.float64, device="cuda")
    op = f.bind_operator([(grad, dev), (div, {"J": dev["J"], "R": dev["R"], "u": other})], 0)
    assert op.entry_points == ("fe_graddiv3d_f64",)


def test_record_facts_measures_and_retrieve_picks_the_faster_variant(tmp_path):
    # the archive loop of examples/howto_autotune.py: record -> query -> retrieve -> timeit
    db = str(tmp_path / "facts.sqlite")
    expr = dg.grad()
    for v in ("generic", "mfma"):
        f.record_facts(expr, 0, v, database=db, long_dim_length=20_000)
    facts = {q.transform_id: q for q in f.query(expr, f.DeviceQueue(0).device, database=db)}
    assert facts["mfma"].runtime_in_sec < facts["generic"].runtime_in_sec
    best = f.retrieve(expr, f.DeviceQueue(0).device, database=db)
    assert dict(best) == {"variant": "mfma", "placement": "split"}        # the fact says how its arrays were placed
    assert facts["mfma"].transform_params == {"placement": "split"}
    assert f.timeit(expr, cq=0, transform=best, long_dim_length=20_000) > 0


def test_operator_capture_and_replay():
    """A multi-launch operator recorded in a HIP graph replays with new input values."""
    import torch

    from oracle import np_oracle

    E = 2003
    exprs = [dg.div(), dg.grad(), dg.face_mass(5), dg.mass_apply(2)]        # four launches (fuse=False)
    hosts = [generate_host_input_arrays(e, E, np_seed=k) for k, e in enumerate(exprs)]
    devs = [{k: torch.from_numpy(v).cuda() for k, v in h.items()} for h in hosts]
    op = f.bind_operator(list(zip(exprs, devs)), 0, fuse=False)
    assert len(op.entry_points) == 4
    with pytest.raises(RuntimeError):
        op.replay()
    op.capture()
    rng = np.random.default_rng(5)
    for h, d in zip(hosts, devs):
        for k in h:
            h[k] = rng.random(h[k].shape)
            d[k].copy_(torch.from_numpy(h[k]))
    for o in op.outputs:
        for t in o.values():
            t.fill_(float("nan"))
    op.replay()
    torch.cuda.synchronize()
    for e, h, o in zip(exprs, hosts, op.outputs):
        for name, row in zip(e.output_names, e.args):
            ref = np_oracle.reference_outputs(e.get_subscripts(), [[h[a.name] for a in row]])[0]
            assert np_oracle.max_rel_err(o[name].cpu().numpy(), ref) <= 1e-12
    assert op.time_batch(3, graph=True) > 0 and op.time_batch(3) > 0


def test_launchers_are_graph_capturable():
    """No allocation / synchronisation inside the launch path: the three family launchers can be
    captured into a HIP graph on a side stream and replayed (inputs updated in place)."""
    import torch

    from oracle import np_oracle

    exprs = [dg.div(), dg.grad(), dg.face_mass()]
    E = 4099
    hosts = [generate_host_input_arrays(e, E, np_seed=k) for k, e in enumerate(exprs)]
    devs = [{k: torch.from_numpy(v).cuda() for k, v in h.items()} for h in hosts]
    outs = [measure.generate_out_arrays(0, e, E) for e in exprs]
    for e, d, o in zip(exprs, devs, outs):          # warm-up outside capture (attribute setup)
        f.evaluate(e, 0, d, out_dict=o)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        for e, d, o in zip(exprs, devs, outs):
            f.evaluate(e, f.DeviceQueue(0, stream=torch.cuda.current_stream()), d, out_dict=o)
    # new input values, same buffers; poison the outputs; replay
    rng = np.random.default_rng(77)
    for h, d in zip(hosts, devs):
        for k in h:
            h[k] = rng.random(h[k].shape)
            d[k].copy_(torch.from_numpy(h[k]))
    for o in outs:
        for t in o.values():
            t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    for e, h, o in zip(exprs, hosts, outs):
        for name, row in zip(e.output_names, e.args):
            ref = np_oracle.reference_outputs(e.get_subscripts(), [[h[a.name] for a in row]])[0]
            assert np_oracle.max_rel_err(o[name].cpu().numpy(), ref) <= 1e-12


def test_dynamic_walk_launches_replay_in_a_graph():
    """Launches whose tiles come by tickets (five or more rounds: feinsum_amd/csrc/fe_common.h, dynamic walk) captured into a
    HIP graph and replayed several times: a launch leaves its ticket counters zeroed, so a replay -- same kernel arguments,
    same counters -- hands out every tile again.  Bitwise the static walk's results for the inputs of each replay."""
    import torch

    from feinsum_amd import _hip

    exprs = [dg.div(), dg.grad(), dg.face_mass(4)]
    E = 200_003
    gen = torch.Generator(device="cuda").manual_seed(3)
    devs = [{a: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in e.arg_to_shape[a]), dtype=torch.float64,
                           device="cuda", generator=gen) for a in sorted(e.all_args)} for e in exprs]
    outs = [measure.generate_out_arrays(0, e, E) for e in exprs]
    for e, d, o in zip(exprs, devs, outs):          # warm-up outside capture (attribute setup, the ticket buffer)
        f.evaluate(e, 0, d, out_dict=o)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        for e, d, o in zip(exprs, devs, outs):
            f.evaluate(e, f.DeviceQueue(0, stream=torch.cuda.current_stream()), d, out_dict=o)
    for replay in range(3):
        for d in devs:
            for t in d.values():
                t.uniform_(0.0, 1.0, generator=gen)
        for o in outs:
            for t in o.values():
                t.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        before = _hip.set_tail_rounds(-1)
        try:
            for e, d, o in zip(exprs, devs, outs):
                static = f.evaluate(e, 0, d, wait=True)
                for name in static:
                    assert torch.equal(static[name], o[name]), (replay, e.get_subscripts(), name)
        finally:
            _hip.set_tail_rounds(before)
