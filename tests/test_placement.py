"""feinsum_amd.placement: arena layout arithmetic (CPU) and the tuned layout on the device."""

import pytest

from feinsum_amd import placement

MIB = placement.MIB


def test_layout_offsets():
    sizes = [3 * MIB + 5, 10, 7 * MIB]
    assert placement.layout_offsets(sizes, 0) == [0, 4 * MIB, 6 * MIB]
    offs = placement.layout_offsets(sizes, 100 * MIB)
    assert offs == [0, 104 * MIB, 206 * MIB]
    assert all(o % placement.ALIGN == 0 for o in offs)
    # arrays never overlap and the arena holds the widest layout
    for gap in (0, MIB, 136 * MIB):
        offs = placement.layout_offsets(sizes, gap)
        assert all(a + s <= b for a, s, b in zip(offs, sizes, offs[1:]))
        assert offs[-1] + sizes[-1] <= placement.arena_bytes(sizes, 136 * MIB)


def test_split_order_puts_one_cut_through_every_stage():
    E = 1000
    stages = [[("0>out", (E, 35), None)],                               # div: one stream
              [("1>out", (3, E, 35), None)],                            # grad: three planes of one array
              [(f"2>out{k}", (E, 35), None) for k in range(4)]]         # face-mass x 4: four arrays
    names = [n for n, _, _ in placement.split_order(stages)]
    assert names == ["0>out", "2>out0", "2>out1", "1>out", "2>out2", "2>out3"]
    # a single stage keeps its order; three outputs split 2 + 1
    assert [n for n, _, _ in placement.split_order([stages[2]])] == [f"2>out{k}" for k in range(4)]
    assert [n for n, _, _ in placement.split_order([stages[2][:3]])] == ["2>out0", "2>out1", "2>out2"]
    assert placement.split_order([]) == []


def test_placement_mode_of_timeit(monkeypatch):
    """``timeit`` times one allocation per array unless asked otherwise (the reference's protocol is the
    default; a tuned arena is opt-in through the transform or the environment)."""
    from feinsum_amd import measure
    from feinsum_amd.diagnostics import InvalidParameterError

    monkeypatch.delenv("FEINSUM_PLACEMENT", raising=False)
    assert measure._placement_mode(None) == "separate" and measure._placement_mode("mfma") == "separate"
    assert measure._placement_mode({"variant": "mfma"}) == "separate"
    assert measure._placement_mode({"placement": "tuned"}) == "tuned"
    assert measure._placement_mode({"placement": "auto"}) == "separate"       # round 2's default name
    monkeypatch.setenv("FEINSUM_PLACEMENT", "tuned")
    assert measure._placement_mode(None) == "tuned"
    assert measure._placement_mode({"placement": "separate"}) == "separate"   # the transform wins over the environment
    with pytest.raises(InvalidParameterError):
        measure._placement_mode({"placement": "somewhere"})
    assert measure.TimingResult(1e-3, 1e-3, 10).placement["mode"] == "separate"


@pytest.mark.gpu
def test_tuned_layout_gives_the_same_results():
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    E, expr = 4099, dg.grad()
    q = f.DeviceQueue(0)
    arrays = [(n, tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), torch.float64)
              for n in sorted(expr.all_args)] + [("_fe_out", (3, E, 35), torch.float64)]

    def fill(name, view):
        if name == "_fe_out":
            view.fill_(float("nan"))
        else:
            view.uniform_(0.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(len(name)))

    def make_step(views):
        _, bound, _ = measure._bind(expr, q, {n: views[n] for n in expr.all_args}, {"_fe_out": views["_fe_out"]}, None)
        return lambda n: bound.time_batch(n, q.stream_ptr)

    arena, views, report = placement.tune_gap(arrays, "cuda", make_step, gaps_mib=(0, 8, 40), fill=fill,
                                              rounds=1, launches=2, warmup=1)
    assert report["best_gap_mib"] in (0, 8, 40) and set(report["ms_by_gap_mib"]) == {"0", "8", "40"}
    assert all(v.data_ptr() % placement.ALIGN == arena.buf.data_ptr() % placement.ALIGN for v in views.values())
    make_step(views)(1)
    q.finish()
    ref = f.evaluate(expr, 0, {n: views[n].clone() for n in expr.all_args}, wait=True)["_fe_out"]
    assert torch.equal(views["_fe_out"], ref)

    # the position scan: same contract, the layout somewhere inside a (here small) arena
    arena, views, report = placement.tune_base(arrays, "cuda", make_step, arena_gib=1.0, gap_mib=2, fill=fill,
                                               coarse_launches=2, launches=2, rounds=1)
    assert report["scan_positions"] >= 2 and report["best_base_mib"] >= 0
    assert arena.buf.numel() >= 0.9 * 2**30
    make_step(views)(1)
    q.finish()
    assert torch.equal(views["_fe_out"], ref)

    # again in fresh arenas while no class boundary shows (a 1 GiB arena has none, as a rule): the fastest is kept
    arena, views, report = placement.tune_base_retry(arrays, "cuda", make_step, attempts=2, arena_gib=1.0, gap_mib=2,
                                                     fill=fill, coarse_launches=2, launches=2, rounds=1)
    assert report["arenas_tried"] in (1, 2) and isinstance(report["class_boundary_found"], bool)
    make_step(views)(1)
    q.finish()
    assert torch.equal(views["_fe_out"], ref)
