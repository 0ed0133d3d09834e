"""
The re-entrancy contract of the C ABI ("distinct streams may be driven from distinct threads", the
executor contract of src/feinsum/measure.py:163-165,243-251) for the one piece of mutable device state the launchers
have -- the ticket counters of the dynamic walk (feinsum_amd/csrc/fe_common.h).  A counter group belongs to a STREAM
(launches on one stream are serialised) or, for a launch recorded during stream capture, to the graph node; launches that
can run at the same time never share one.  Two launches drawing tickets from one counter would each skip the tiles the
other took -- silently -- so every check here is bitwise against the static walk, which needs no state.
"""

import threading

import pytest

import dg
import feinsum_amd as f
from feinsum_amd import _hip, measure

E = 200_000        # 12 500 tiles on 2048 waves: six rounds, four of them by tickets


def _inputs(torch, expr, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return {a: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[a]),
                          dtype=torch.float64, device="cuda", generator=g) for a in sorted(expr.all_args)}


@pytest.fixture(scope="module")
def cases():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    exprs = {"grad": dg.grad(), "div": dg.div(), "face_mass": dg.face_mass(4)}
    devs = {k: _inputs(torch, e, 11 + i) for i, (k, e) in enumerate(exprs.items())}
    before = _hip.set_tail_rounds(-1)          # the static walk: the reference bits
    try:
        static = {k: {n: t.clone() for n, t in f.evaluate(exprs[k], 0, devs[k], wait=True).items()} for k in exprs}
    finally:
        _hip.set_tail_rounds(before)
    return torch, exprs, devs, static


def _outputs(torch, expr):
    shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    return {n: torch.full(shape, float("nan"), dtype=torch.float64, device="cuda") for n in expr.output_names}


@pytest.mark.gpu
def test_two_threads_two_streams_forty_launches_each(cases):
    """Two host threads, one stream each, forty alternating grad / div / face-mass launches per thread, nothing waited for
    in between: more launches in flight than the sixteen counter groups round 3 handed out in turn (launch k and launch
    k + 16 then shared counters).  Every launch writes its own output arrays; all of them hold the static walk's bits."""
    torch, exprs, devs, static = cases
    names = list(exprs)
    results, errors = {}, []
    start = threading.Barrier(2)

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            q = f.DeviceQueue(0, stream=stream)
            bound = []
            for k in range(40):
                name = names[(k + tid) % 3]
                outs = _outputs(torch, exprs[name])
                bound.append((name, outs, measure._bind(exprs[name], q, devs[name], outs, None)[1]))
            stream.wait_stream(torch.cuda.current_stream())      # (the NaN fills ran on the thread's current stream)
            start.wait()
            with torch.cuda.device(0):
                for _, _, b in bound:
                    b.launch(q.stream_ptr)
            stream.synchronize()
            results[tid] = [(name, outs) for name, outs, _ in bound]
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)
            start.abort()

    before = _hip.set_tail_rounds(1 << 20)
    try:
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        _hip.set_tail_rounds(before)
    assert not errors, errors
    for tid in (0, 1):
        for k, (name, outs) in enumerate(results[tid]):
            for n, t in outs.items():
                assert torch.equal(t, static[name][n]), (tid, k, name, n)
    check = _hip.tail_check()
    assert check["dirty_words"] == 0 and check["streams"] >= 2, check


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_graph_replays_beside_eager_launches(cases, fuse):
    """A captured operator (div + grad + face-mass x 4: one fused launch with a counter set per body, or three launches)
    replayed on stream A while eager launches of the same families run on stream B: the graph's launches own their
    counters for good, so neither side disturbs the other."""
    torch, exprs, devs, static = cases
    shared = dict(devs["grad"])                                    # div and grad of one operator share J and D
    ddev = dict(devs["div"], J=shared["J"], R=shared["R"])
    stages = [(exprs["div"], ddev), (exprs["grad"], shared), (exprs["face_mass"], devs["face_mass"])]
    before = _hip.set_tail_rounds(-1)
    try:
        ref = [{n: t.clone() for n, t in od.items()} for od in f.evaluate_operator(stages, 0, fuse=fuse, wait=True)]
        _hip.set_tail_rounds(1 << 20)
        outs_graph = [_outputs(torch, e) for e, _ in stages]
        op = f.bind_operator(stages, 0, out_dicts=outs_graph, fuse=fuse)
        assert len(op.launches) == (1 if fuse else 3)
        op.capture()
        for od in outs_graph:                                      # (capture() ran the operator once: start from NaN again)
            for t in od.values():
                t.fill_(float("nan"))
        torch.cuda.synchronize()
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        qb = f.DeviceQueue(0, stream=b)
        eager = []
        for k in range(24):
            name = list(exprs)[k % 3]
            outs = _outputs(torch, exprs[name])
            eager.append((name, outs, measure._bind(exprs[name], qb, devs[name], outs, None)[1]))
        torch.cuda.synchronize()
        for rep in range(8):
            with torch.cuda.stream(a):
                op.replay()
            with torch.cuda.device(0):
                for name, outs, bnd in eager[3 * rep:3 * rep + 3]:
                    bnd.launch(qb.stream_ptr)
        torch.cuda.synchronize()
    finally:
        _hip.set_tail_rounds(before)
    for k, od in enumerate(outs_graph):
        for n, t in od.items():
            assert torch.equal(t, ref[k][n]), ("graph", fuse, k, n)
    for k, (name, outs, _) in enumerate(eager):
        for n, t in outs.items():
            assert torch.equal(t, static[name][n]), ("eager", fuse, k, name, n)
    check = _hip.tail_check()
    assert check["dirty_words"] == 0 and check["captured"] >= 1, check


@pytest.mark.gpu
def test_a_stream_keeps_its_group_and_a_retired_stream_gives_it_back(cases):
    torch, exprs, devs, static = cases
    before = _hip.set_tail_rounds(1 << 20)
    try:
        s1 = torch.cuda.Stream()
        q1 = f.DeviceQueue(0, stream=s1)
        n0 = _hip.tail_check()["streams"]
        for _ in range(3):
            f.evaluate(exprs["grad"], q1, devs["grad"], wait=True)
        assert _hip.tail_check()["streams"] == n0 + 1                 # one group per stream, however many launches
        assert _hip.stream_retired(int(s1.cuda_stream)) is True
        assert _hip.stream_retired(int(s1.cuda_stream)) is False
        assert _hip.tail_check()["streams"] == n0
        out = f.evaluate(exprs["grad"], q1, devs["grad"], wait=True)["_fe_out"]     # ... and takes one again when it launches
        assert torch.equal(out, static["grad"]["_fe_out"])
    finally:
        _hip.set_tail_rounds(before)


@pytest.mark.gpu
@pytest.mark.parametrize("cus", [32, 64, 100])
def test_small_grids_walk_statically(cases, cus):
    """A 32-CU (CPX) or 64-CU (QPX) partition of MI355X launches 64 / 128 blocks -- half of that for the eight-wave p = 5
    kernels: below 128 blocks some of the sixteen ticket pools have no block to drain them (a block's pool is (bid / 8)
    mod 16, nobody steals), and round 3 left their tiles unwritten.  fe_set_cu_limit sizes the grids as such a
    device would; outputs are pre-filled with NaN; the reference bits are those of the full grid."""
    torch, exprs, devs, static = cases
    more = {"grad_p5": dg.grad(56), "face_mass_p5": dg.face_mass(4, 56, 4, 21)}
    mdev = {k: _inputs(torch, e, 31 + i) for i, (k, e) in enumerate(more.items())}
    before_rounds = _hip.set_tail_rounds(1 << 20)
    try:
        full = {k: {n: t.clone() for n, t in f.evaluate(more[k], 0, mdev[k], wait=True).items()} for k in more}
        before_cus = _hip.set_cu_limit(cus)
        try:
            for name, expr in list(exprs.items()) + list(more.items()):
                outs = _outputs(torch, expr)
                f.evaluate(expr, 0, (devs if name in devs else mdev)[name], out_dict=outs, wait=True)
                for n, t in outs.items():
                    assert torch.equal(t, (static if name in static else full)[name][n]), (cus, name, n)
        finally:
            assert _hip.set_cu_limit(before_cus) == cus
    finally:
        _hip.set_tail_rounds(before_rounds)
    assert _hip.tail_check()["dirty_words"] == 0


@pytest.mark.gpu
def test_tail_check_finds_and_repairs_a_stale_counter(cases):
    """The invariant every dynamic launch relies on -- counters are zero between launches -- is checkable and repairable
    (a launch that did not run to completion would leave tickets behind, and later launches through that group would skip
    tiles: ADVICE r03).  A stale ticket is planted by hand in the default stream's group."""
    torch, exprs, devs, static = cases
    before = _hip.set_tail_rounds(1 << 20)
    try:
        f.evaluate(exprs["grad"], 0, devs["grad"], wait=True)
        assert _hip.tail_check()["dirty_words"] == 0
        _hip.tail_plant(int(torch.cuda.current_stream().cuda_stream), 5)
        assert _hip.tail_check()["dirty_words"] == 1
        assert _hip.tail_check(repair=True)["dirty_words"] == 1      # reports what it found, then zeroes it
        assert _hip.tail_check()["dirty_words"] == 0
        out = f.evaluate(exprs["grad"], 0, devs["grad"], wait=True)["_fe_out"]
        assert torch.equal(out, static["grad"]["_fe_out"])
    finally:
        _hip.set_tail_rounds(before)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["grad", "div", "face_mass", "pipeline"])
def test_plain_and_non_temporal_loads_give_the_same_bits(cases, name):
    """Launches whose inputs fit the Infinity Cache fetch their streamed operand with plain instead of non-temporal loads
    (fe_set_temporal_loads_mib; feinsum_amd/csrc/fe_common.h): a cache hint, not arithmetic -- the same bits either way, for
    the single launches and for the fused operator (one launch, three bodies)."""
    torch, exprs, devs, static = cases
    if name == "pipeline":
        shared = dict(devs["grad"])
        stages = [(exprs["div"], dict(devs["div"], J=shared["J"], R=shared["R"])), (exprs["grad"], shared),
                  (exprs["face_mass"], devs["face_mass"])]
    else:
        stages = [(exprs[name], devs[name])]
    before = _hip.set_temporal_loads_mib(0)
    try:
        results = []
        for mib in (0, 1 << 20, 248):
            _hip.set_temporal_loads_mib(mib)
            results.append([{n: t.clone() for n, t in od.items()} for od in f.evaluate_operator(stages, 0, wait=True)])
        for other in results[1:]:
            for od, rd in zip(other, results[0]):
                for n in rd:
                    assert torch.equal(od[n], rd[n]), (name, n)
        if name != "pipeline":
            for n, t in results[0][0].items():
                assert torch.equal(t, static[name][n])
    finally:
        assert _hip.set_temporal_loads_mib(before) == 248


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["grad", "grad_p3", "grad_p2", "grad_p1", "div", "div_p3", "face_mass", "face_mass_b2", "face_mass_p2", "grad_2d"])
def test_write_through_and_non_temporal_stores_give_the_same_bits(cases, name):
    """Short grad launches (static walk, outputs of at most fe_set_write_through_mib MiB) store write-through instead of
    non-temporally (feinsum_amd/csrc/fe_common.h): a cache policy, not arithmetic -- the same bits either way, at sizes with
    one tile per wave, several, a partial last round and elements behind the last tile; every byte of the outputs is written
    (NaN-filled first).  The other families never take the flag and must be unaffected by the setting."""
    torch, *_ = cases
    expr = {"grad": lambda: dg.grad(), "grad_p3": lambda: dg.grad(20), "grad_p2": lambda: dg.grad(10), "grad_p1": lambda: dg.grad(4),
            "div": lambda: dg.div(), "div_p3": lambda: dg.div(20), "face_mass": lambda: dg.face_mass(4), "face_mass_b2": lambda: dg.face_mass(2),
            "face_mass_p2": lambda: dg.face_mass(4, 10, 4, 6),
            "grad_2d": lambda: f.einsum("xre,rij,ej->xei", f.array("J", (2, 2, "E")), f.array("R", (2, 15, 15)), f.array("u", ("E", 15)))}[name]()
    before = _hip.set_write_through_mib(0)
    try:
        for E in (16, 1000, 30_001, 100_000, 100_003, 150_000):
            g = torch.Generator(device="cuda").manual_seed(E)
            dev = {a: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[a]), dtype=torch.float64,
                                 device="cuda", generator=g) for a in sorted(expr.all_args)}
            shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
            results = []
            for mib in (0, 1 << 20):
                _hip.set_write_through_mib(mib)
                outs = {n: torch.full(shape, float("nan"), dtype=torch.float64, device="cuda") for n in expr.output_names}
                f.evaluate(expr, 0, dev, out_dict=outs, wait=True)
                results.append(outs)
            for n in expr.output_names:
                assert not torch.isnan(results[1][n]).any(), (name, E, n)
                assert torch.equal(results[0][n], results[1][n]), (name, E, n)
    finally:
        assert _hip.set_write_through_mib(before) == 1 << 20


@pytest.mark.gpu
def test_a_recaptured_operator_gives_its_groups_back(cases):
    """Round 5: the counter groups a capture's launches own come back when the graph is destroyed
    (fe_graph_retired through BoundOperator.release_graph), so an application that captures again every step never runs out of
    groups; `exhausted` / `static_fallbacks` of fe_tail_stats would say so if it did.  300 captures > the 256 groups a device
    may hold; every replay bitwise the static walk."""
    torch, exprs, devs, static = cases
    before = _hip.set_tail_rounds(1 << 20)
    try:
        outs = _outputs(torch, exprs["grad"])
        op = f.bind_operator([(exprs["grad"], devs["grad"])], 0, out_dicts=[outs])
        s0 = _hip.tail_stats()
        op.capture()
        s1 = _hip.tail_stats()
        assert s1["captured"] == s0["captured"] + 1 and s1["live_captures"] == s0["live_captures"] + 1
        for rep in range(300):
            op.capture()                      # releases the previous graph's group first
            if rep % 50 == 0:
                outs["_fe_out"].fill_(float("nan"))
                op.replay()
                torch.cuda.synchronize()
                assert torch.equal(outs["_fe_out"], static["grad"]["_fe_out"]), rep
        s2 = _hip.tail_stats()
        assert s2["captured"] == s1["captured"] and s2["exhausted"] == s0["exhausted"] and s2["static_fallbacks"] == s0["static_fallbacks"], (s0, s2)
        assert op.release_graph() == 1 and op.release_graph() == 0
        s3 = _hip.tail_stats()
        assert s3["captured"] == s0["captured"] and s3["live_captures"] == s0["live_captures"]
        del op
    finally:
        _hip.set_tail_rounds(before)
    assert _hip.tail_check()["dirty_words"] == 0


@pytest.mark.gpu
def test_after_a_hip_error_the_first_dynamic_launch_verifies_its_group(cases):
    """Round 5: the zero-between-launches invariant is no longer trusted blindly after an error.
    Any FE_EHIP return of the process makes every stream's next dynamic launch wait for the stream once and verify (repair) its
    counter group.  Here: a stale ticket planted by hand, then an FE_EHIP return (properties of a device that does not exist) --
    the next launch finds and repairs the ticket and computes the right bits; without the error nothing is checked."""
    torch, exprs, devs, static = cases
    import ctypes as C

    before = _hip.set_tail_rounds(1 << 20)
    try:
        f.evaluate(exprs["grad"], 0, devs["grad"], wait=True)          # the default stream owns a group
        s0 = _hip.tail_stats()
        lib = _hip.load_library()
        assert lib.fe_device_info(99, None, 0, None, None) == -3        # FE_EHIP: no such device (not a sticky error)
        assert _hip.tail_stats()["hip_errors"] == s0["hip_errors"] + 1
        _hip.tail_plant(int(torch.cuda.current_stream().cuda_stream), 7)
        out = f.evaluate(exprs["grad"], 0, devs["grad"], wait=True)["_fe_out"]
        s1 = _hip.tail_stats()
        assert s1["verified_after_error"] == s0["verified_after_error"] + 1 and s1["repaired_after_error"] == s0["repaired_after_error"] + 1
        assert torch.equal(out, static["grad"]["_fe_out"])
        out = f.evaluate(exprs["grad"], 0, devs["grad"], wait=True)["_fe_out"]     # verified once, not again
        assert _hip.tail_stats()["verified_after_error"] == s1["verified_after_error"]
        assert torch.equal(out, static["grad"]["_fe_out"])
    finally:
        _hip.set_tail_rounds(before)
    assert _hip.tail_check()["dirty_words"] == 0
