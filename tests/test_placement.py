"""feinsum_amd.placement: the placement modes of timeit / evaluate (CPU) and the split allocator on the device."""

from pathlib import Path

import numpy as np
import pytest

from feinsum_amd import placement

ROOT = Path(__file__).resolve().parents[1]

MIB = placement.MIB


def test_placement_mode_of_timeit(monkeypatch):
    """``timeit`` times one allocation per array (outputs from the split allocator) unless asked otherwise; round 2's
    arena scan ("tuned") is gone."""
    from feinsum_amd import measure
    from feinsum_amd.diagnostics import InvalidParameterError

    monkeypatch.delenv("FEINSUM_PLACEMENT", raising=False)
    assert measure._placement_mode(None) == "split" and measure._placement_mode("mfma") == "split"
    assert measure._placement_mode({"variant": "mfma"}) == "split"
    assert measure._placement_mode({"placement": "separate"}) == "separate"
    assert measure._placement_mode({"placement": "auto"}) == "split"          # round 2's default name
    monkeypatch.setenv("FEINSUM_PLACEMENT", "separate")
    assert measure._placement_mode(None) == "separate"
    assert measure._placement_mode({"placement": "split"}) == "split"         # the transform wins over the environment
    for gone in ("somewhere", "tuned"):
        with pytest.raises(InvalidParameterError):
            measure._placement_mode({"placement": gone})
    assert measure.TimingResult(1e-3, 1e-3, 10).placement["mode"] == "separate"


def test_outputs_evaluate_allocates_itself(monkeypatch):
    """measure._allocate_output: what `evaluate` does for an output it is not handed -- small arrays, CPU devices and
    `placement: separate` are plain torch allocations; large device arrays ask the split allocator and fall back to torch when
    it cannot serve (here: no device at all), saying which it was."""
    import torch

    from feinsum_amd import measure

    monkeypatch.delenv("FEINSUM_PLACEMENT", raising=False)
    t, how = measure._allocate_output((1000, 35), torch.float64, "cpu", None)
    assert how == "torch" and t.shape == (1000, 35) and t.dtype == torch.float64
    t, how = measure._allocate_output((0, 35), torch.float64, "cpu", None)
    assert how == "torch" and t.numel() == 0


@pytest.mark.gpu
def test_evaluate_falls_back_to_torch_when_the_allocator_cannot_serve(monkeypatch):
    """... and on a device: the allocator for 8 MiB and more, torch when it fails or when asked (`output_allocations` of the
    bound launch says which)."""
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    monkeypatch.delenv("FEINSUM_PLACEMENT", raising=False)
    E, expr = 100_000, dg.grad()
    dev = {k: torch.from_numpy(v).cuda() for k, v in measure.generate_host_input_arrays(expr, E).items()}
    _, bound, outs = measure._bind(expr, 0, dev, None, None)
    assert dict(bound.output_allocations) == {"_fe_out": "split"} and placement.split_info(outs[0])["pieces"] > 0
    _, bound, outs = measure._bind(expr, 0, dev, None, {"placement": "separate"})
    assert dict(bound.output_allocations) == {"_fe_out": "torch (placement: separate)"} and placement.split_info(outs[0]) == {}

    def no_vmm(*a, **k):
        raise RuntimeError("no VMM here")

    monkeypatch.setattr(placement, "empty", no_vmm)
    _, bound, outs = measure._bind(expr, 0, dev, None, None)
    assert bound.output_allocations["_fe_out"].startswith("torch (split allocator failed") and outs[0].shape == (3, E, 35)
    ref = f.evaluate(expr, 0, dev, out_dict={"_fe_out": torch.empty_like(outs[0])}, wait=True)["_fe_out"]
    bound.launch(0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], ref)
    _, bound, _ = measure._bind(dg.grad(), 0, {k: torch.from_numpy(v).cuda() for k, v in measure.generate_host_input_arrays(expr, 1000).items()},
                                None, None)
    assert dict(bound.output_allocations) == {"_fe_out": "torch"}              # below 8 MiB


@pytest.mark.gpu
def test_split_allocator_arrays_are_ordinary_tensors_with_the_same_results():
    """feinsum_amd.placement.empty: arrays of the split allocator (fe_split_alloc) behind torch tensors -- results of
    launches that write into them are bitwise those of torch allocations; the allocator's report; memory comes back."""
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import _hip

    E = 1_000_000
    placement.recycle_trim(0)          # (arrays of earlier tests that wait for reuse: really freed, so that the counts below are this test's)
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
    ref = f.evaluate(expr, 0, dev, transform={"placement": "separate"}, wait=True)["_fe_out"]
    assert placement.split_info(ref) == {}                        # a plain torch allocation, as asked
    own = f.evaluate(expr, 0, dev, wait=True)["_fe_out"]          # an output evaluate() allocates itself: from the allocator
    assert placement.split_info(own)["pieces"] == 3 * E * 35 * 8 // (4 << 20) and torch.equal(own, ref)
    del own
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
    fref = f.evaluate(fm, 0, fdev, transform={"placement": "separate"}, wait=True)
    fres = f.evaluate(fm, 0, fdev, out_dict=dict(zip(fm.output_names, outs)), wait=True)
    for name in fm.output_names:
        assert torch.equal(fres[name], fref[name])
    # small arrays, read-only arrays and CPU arrays are plain torch allocations
    assert placement.split_info(placement.empty((1000, 35), torch.float64, "cuda:0")) == {}
    assert placement.split_info(placement.empty((3, E, 35), torch.float64, "cuda:0", written=False)) == {}
    assert placement.empty((10, 3), torch.float32, "cpu").device.type == "cpu"
    # an array whose last view is gone waits for reuse (round 5: no unmap, no device synchronisation on release) ...
    kept = placement.recycle_stats()["kept"]
    view = out[2]
    del out, res
    assert placement.recycle_stats()["kept"] == kept              # the view keeps the array
    del view
    assert placement.recycle_stats()["kept"] == kept + 1
    again = placement.empty((3, E, 35), torch.float64, "cuda:0")  # ... and the next array of that size IS it
    assert placement.recycle_stats()["reused"] >= 1 and placement.split_info(again)["pieces"] == n
    del again, outs, fres
    # ... and the memory returns to the pool when the waiting arrays are trimmed
    placement.recycle_trim(0)
    after = placement.split_stats(0)
    assert after["live_arrays"] == before["live_arrays"] and after["live_bytes"] == before["live_bytes"]
    assert placement.recycle_stats()["free_failures"] == 0
    with pytest.raises(f.InvalidParameterError):
        _hip.split_free(12345 * 4096)                             # not an array of the allocator


@pytest.mark.gpu
def test_split_allocator_never_hands_out_an_address_twice(monkeypatch):
    """ROCm 7.2 keeps translating a re-mapped virtual range to its FIRST physical handle (tools/vmm_remap_test.cpp), so
    the allocator must never re-use an address: allocate / free cycles return distinct pointers, and what is written
    through a new array is what is read back after the pool has recycled the physical pieces.  (With the Python-side
    recycling of whole arrays switched off -- a recycled array keeps its mapping, which is the point of recycling.)"""
    import torch

    monkeypatch.setenv("FEINSUM_SPLIT_RECYCLE_MIB", "0")
    placement.recycle_trim(0)
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
    torch.cuda.empty_cache()           # (the suite's earlier tests leave tens of GB in torch's cache: less room for the search)
    placement.split_trim("cuda:0")
    placement.split_reserve(total, "cuda:0")
    pool = placement.split_stats("cuda:0")
    if pool["walk_gave_up"]:
        # the search for a second class is bounded (96 GiB skipped or 4 s) and where the driver's classes lie is not ours to
        # choose: in such a process the arrays are of one class and say so -- nothing to check about the announcement
        pytest.skip("no second class of physical memory within the allocator's search budget in this process")
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


@pytest.mark.gpu
def test_evaluate_without_out_dict_recycles_its_outputs():
    """an ``evaluate()`` that allocates its own outputs must not pay a VMM map, an unmap and a device
    synchronisation per call.  The output of the previous call, dropped by the caller, is the output of the next one (same
    address, stream-ordered through an event); results stay right; inside a stream capture the output is torch's."""
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    E = 400_000
    expr = dg.grad()
    g = torch.Generator(device="cuda").manual_seed(9)
    dev = {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64,
                         device="cuda", generator=g) for n in sorted(expr.all_args)}
    ref = f.evaluate(expr, 0, dev, transform={"placement": "separate"}, wait=True)["_fe_out"]
    placement.recycle_trim(0)
    s0 = placement.recycle_stats()
    ptrs = set()
    for k in range(20):
        out = f.evaluate(expr, 0, dev)["_fe_out"]          # asynchronous; the previous output is dropped here
        ptrs.add(out.data_ptr())
        if k % 5 == 4:
            assert torch.equal(out, ref)
        if k == 4:                                         # (the first calls map their two arrays and the pool's groups)
            va_early = placement.split_stats(0)["address_space_reserved"]
    torch.cuda.synchronize()
    s1 = placement.recycle_stats()
    # (`out` of call k is still alive while call k + 1 allocates: two arrays take turns)
    assert len(ptrs) <= 2 and s1["reused"] - s0["reused"] >= 18, (ptrs, s0, s1)
    assert placement.split_stats(0)["address_space_reserved"] == va_early      # no address space per call any more
    # inside a capture the allocator is not touched
    q = f.DeviceQueue(0)
    _, bound, outs = measure._bind(expr, q, dev, None, None)
    assert bound.output_allocations["_fe_out"] == "split"
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        _, bound_c, outs_c = measure._bind(expr, f.DeviceQueue(0, stream=torch.cuda.current_stream()), dev, None, None)
        bound_c.launch(int(torch.cuda.current_stream().cuda_stream))
    assert "capture" in bound_c.output_allocations["_fe_out"]
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs_c[0], ref)
    del out, outs, outs_c, bound, bound_c, graph
    placement.recycle_trim(0)


@pytest.mark.gpu
def test_an_array_of_one_class_is_refused_and_allocated_ordinarily(tmp_path):
    """Round 5: when the allocator finds no second class of physical memory within its budget it REFUSES the array (an array
    whose pieces are all of one class is the worst placement there is: every write stream in one class) and
    ``placement.empty`` allocates ordinarily -- ``evaluate`` keeps working and says what it got.  ``FEINSUM_SPLIT_UNSPLIT=1``
    restores the array of one class.  In a fresh process each (the pool reads its knobs once; ``FEINSUM_SPLIT_ONE_CLASS=1``
    makes it behave as if the device offered one class)."""
    import json
    import os
    import subprocess
    import sys

    code = r"""
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import dg
import feinsum_amd as f
from feinsum_amd import placement
t = placement.empty((3, 200_000, 35), torch.float64, "cuda:0")
stats = placement.split_stats("cuda:0")
expr = dg.grad()
E = 200_000
g = torch.Generator(device="cuda").manual_seed(1)
dev = {n: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]), dtype=torch.float64, device="cuda", generator=g)
       for n in sorted(expr.all_args)}
out = f.evaluate(expr, 0, dev, wait=True)["_fe_out"]
ref = f.evaluate(expr, 0, dev, wait=True, transform={"placement": "separate"})["_fe_out"]
print(json.dumps({"is_split": placement.is_split(t), "info": placement.split_info(t), "refused": stats["unsplit_refused"], "unsplit": stats["unsplit_arrays"],
                  "fallbacks": placement.ordinary_fallbacks(), "same": bool(torch.equal(out, ref)), "out_is_split": placement.is_split(out)}))
""" % (str(ROOT), str(ROOT / "tests"))
    results = {}
    for unsplit in ("0", "1"):
        env = dict(os.environ, FEINSUM_SPLIT_ONE_CLASS="1", FEINSUM_SPLIT_UNSPLIT=unsplit, FEINSUM_SPLIT_SEARCH_MS="300", FEINSUM_SPLIT_SEARCH_GIB="2")
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        results[unsplit] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    refused, handed = results["0"], results["1"]
    assert not refused["is_split"] and refused["refused"] >= 1 and refused["unsplit"] == 0 and refused["fallbacks"] >= 1, refused
    assert refused["same"] and not refused["out_is_split"], refused
    assert handed["is_split"] and handed["unsplit"] >= 1 and handed["refused"] == 0 and handed["same"], handed
    by_class = [c for c in handed["info"]["pieces_by_class"] if c]
    assert len(by_class) == 1, handed["info"]
