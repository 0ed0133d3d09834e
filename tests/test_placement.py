"""feinsum_amd.placement: arena layout arithmetic (CPU) and the tuned layout on the device."""

import numpy as np
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
    """``timeit`` times one allocation per array (outputs from the split allocator) unless asked otherwise (ADVICE r02:
    the tuned arena is opt-in through the transform or the environment)."""
    from feinsum_amd import measure
    from feinsum_amd.diagnostics import InvalidParameterError

    monkeypatch.delenv("FEINSUM_PLACEMENT", raising=False)
    assert measure._placement_mode(None) == "split" and measure._placement_mode("mfma") == "split"
    assert measure._placement_mode({"variant": "mfma"}) == "split"
    assert measure._placement_mode({"placement": "tuned"}) == "tuned"
    assert measure._placement_mode({"placement": "separate"}) == "separate"
    assert measure._placement_mode({"placement": "auto"}) == "split"          # round 2's default name
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


@pytest.mark.gpu
def test_split_allocator_arrays_are_ordinary_tensors_with_the_same_results():
    """feinsum_amd.placement.empty: arrays of the split allocator (fe_split_alloc) behind torch tensors -- results of
    launches that write into them are bitwise those of torch allocations; the allocator's report; memory comes back."""
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import _hip

    E = 1_000_000
    before = placement.split_stats(0)
    out = placement.empty((3, E, 35), torch.float64, "cuda:0")
    assert out.shape == (3, E, 35) and out.dtype == torch.float64 and out.is_contiguous() and out.device.index == 0
    info = placement.split_info(out)
    assert info["bytes"] == 3 * E * 35 * 8 and info["bytes"] <= info["mapped_bytes"] < info["bytes"] + (2 << 20)
    n = 3 * E * 35 * 8 // (4 << 20)
    assert info["piece_mib"] == 4 and info["pieces"] == n and sum(info["pieces_by_class"]) == n
    assert info["tail_bytes"] == info["mapped_bytes"] - n * (4 << 20)
    stats = placement.split_stats(0)
    if stats["classes"] >= 2 and not stats["unsplit_arrays"]:     # (a device of one class has nothing to alternate)
        used = sorted(c for c in info["pieces_by_class"] if c)
        assert len(used) == 2 and used[1] - used[0] <= 1, info      # two classes, equal shares ...
        head = info["first_pieces"]
        assert len(head) == 16 and all(head[k] != head[k + 1] for k in range(15)), head   # ... alternating piece by piece
    assert placement.split_info(out[1]) == info                   # views belong to the same array
    assert placement.split_info(torch.empty(4, device="cuda:0")) == {}
    # torch kernels and feinsum launches on the array
    out.fill_(float("nan"))
    expr = dg.grad()
    g = torch.Generator(device="cuda").manual_seed(5)
    dev = {"J": torch.rand((3, 3, E), dtype=torch.float64, device="cuda", generator=g),
           "R": torch.rand((3, 35, 35), dtype=torch.float64, device="cuda", generator=g),
           "u": torch.rand((E, 35), dtype=torch.float64, device="cuda", generator=g)}
    ref = f.evaluate(expr, 0, dev, wait=True)["_fe_out"]
    res = f.evaluate(expr, 0, dev, out_dict={"_fe_out": out}, wait=True)["_fe_out"]
    assert res.data_ptr() == out.data_ptr() and torch.equal(res, ref)
    assert float(out.sum()) == float(ref.sum())
    # four face-mass outputs allocated one after the other: the orientation alternates
    outs = [placement.zeros((E, 35), torch.float64, "cuda:0") for _ in range(4)]
    infos = [placement.split_info(t) for t in outs]
    assert all(i["pieces"] == E * 35 * 8 // (4 << 20) for i in infos)
    if stats["classes"] >= 2 and not placement.split_stats(0)["unsplit_arrays"]:
        assert all(i["first_pieces"][0] != i["first_pieces"][1] for i in infos), [i["first_pieces"] for i in infos]
    fm = dg.face_mass(4)
    fdev = {name: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in fm.arg_to_shape[name]),
                             dtype=torch.float64, device="cuda", generator=g) for name in sorted(fm.all_args)}
    fref = f.evaluate(fm, 0, fdev, wait=True)
    fres = f.evaluate(fm, 0, fdev, out_dict=dict(zip(fm.output_names, outs)), wait=True)
    for name in fm.output_names:
        assert torch.equal(fres[name], fref[name])
    # small arrays, read-only arrays and CPU arrays are plain torch allocations
    assert placement.split_info(placement.empty((1000, 35), torch.float64, "cuda:0")) == {}
    assert placement.split_info(placement.empty((3, E, 35), torch.float64, "cuda:0", written=False)) == {}
    assert placement.empty((10, 3), torch.float32, "cpu").device.type == "cpu"
    # the memory returns to the pool with the last view
    live = placement.split_stats(0)["live_arrays"]
    view = out[2]
    del out, res
    assert placement.split_stats(0)["live_arrays"] == live        # the view keeps the array
    del view
    assert placement.split_stats(0)["live_arrays"] == live - 1
    del outs, fres
    after = placement.split_stats(0)
    assert after["live_arrays"] == before["live_arrays"] and after["live_bytes"] == before["live_bytes"]
    with pytest.raises(f.InvalidParameterError):
        _hip.split_free(12345 * 4096)                             # not an array of the allocator


@pytest.mark.gpu
def test_split_allocator_never_hands_out_an_address_twice():
    """ROCm 7.2 keeps translating a re-mapped virtual range to its FIRST physical handle (tools/vmm_remap_test.cpp), so
    the allocator must never re-use an address: allocate / free cycles return distinct pointers, and what is written
    through a new array is what is read back after the pool has recycled the physical pieces."""
    import torch

    seen = set()
    for cycle in range(6):
        t = placement.empty((40_000_000,), torch.float64, "cuda:0")          # 320 MB: 76 pieces and a tail
        assert t.data_ptr() not in seen
        seen.add(t.data_ptr())
        t.fill_(float(cycle + 1))
        u = placement.empty((40_000_000,), torch.float64, "cuda:0")
        u.fill_(-float(cycle + 1))
        assert float(t[0]) == cycle + 1 and float(t[-1]) == cycle + 1 and float(t.sum()) == (cycle + 1) * 40_000_000.0
        assert float(u[0]) == -(cycle + 1) and float(u[-1]) == -(cycle + 1)
        del t, u


@pytest.mark.gpu
def test_split_reserve_collects_both_classes_for_arrays_allocated_one_after_the_other():
    """``placement.split_reserve(total)`` announces what is about to be allocated: afterwards the pool holds at least half of
    that of each of two classes, and every one of the arrays allocated one after the other is split between the two (without
    the announcement the first array could use up the first class the allocator's search had collected: bench.py's pipeline
    once left its four lift outputs unsplit)."""
    import torch

    shapes = [(3, 1_000_000, 35), (1_000_000, 35)] + [(1_000_000, 35)] * 4        # the pipeline's outputs: 2.24 GB
    total = sum(8 * int(np.prod(sh)) for sh in shapes)
    placement.split_trim("cuda:0")
    placement.split_reserve(total, "cuda:0")
    pool = placement.split_stats("cuda:0")
    free = sorted(pool["free_pieces"], reverse=True) if "free_pieces" in pool else None
    arrays = [placement.empty(sh, torch.float64, "cuda:0") for sh in shapes]
    for t in arrays:
        info = placement.split_info(t)
        by_class = sorted(info["pieces_by_class"], reverse=True)
        assert by_class[0] > 0 and by_class[1] > 0 and abs(by_class[0] - by_class[1]) <= 1, info["pieces_by_class"]
    after = placement.split_stats("cuda:0")
    assert not after["walk_gave_up"] and after["unsplit_arrays"] == pool["unsplit_arrays"]
    if free is not None:
        assert free[1] * placement.split_info(arrays[0])["piece_mib"] * (1 << 20) >= total // 2 - (64 << 20)
    del arrays
