"""
Where the operand arrays of a launch sit in device memory.

New functionality (the reference allocates every array separately through PyOpenCL,
``src/feinsum/measure.py:44-60,80-108``, and has no notion of placement).  On MI355X the same
launch on the same device runs up to 14 % apart depending on where its arrays lie in PHYSICAL
memory (``profiles/r02/placement_*.txt``: face-mass x 4 at E = 1e6 0.485 ... 0.569 ms, grad 0.191
... 0.215 ms): a DG launch streams 13 (grad) to 26 (face-mass) arrays and array slabs in lockstep,
and how those streams fall onto the memory channels and DRAM banks follows from their physical
address bits (the channel hash folds in bits up to the GiB range).  Measured: a layout that lies
inside one physically contiguous block of the driver's allocator is "slow" wherever it is put and
however its arrays are spaced; the same layout laid ACROSS the joint of two such blocks (a large
allocation is built from power-of-two blocks: 64 + 64 + 32 + 16 ... GiB) is 11-14 % faster,
reproducibly.  Separate allocations land wherever the allocator puts them, which is what made the
round-1 numbers differ "between devices".

:class:`Arena` carves all arrays of a workload out of ONE large allocation; :func:`tune_base`
moves the layout through the arena, times the bound launch at every position and keeps the
fastest (it finds the joints); :func:`tune_gap` does the same over the spacing of the arrays.
Both are autotuning steps in the spirit of the reference's transform search
(``src/feinsum/tuning/__init__.py:573-633``), over memory layout instead of loop structure.
The kernels and their results do not depend on placement.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, List, Sequence, Tuple

MIB = 1 << 20


# --------------------------------------------------------------------------
# the split allocator (round 3): arrays whose 4 MiB pieces alternate between
# two classes of physical memory -- no arena, no timing scan
# (include/feinsum_hip.h, feinsum_amd/csrc/fe_split_alloc.h)
# --------------------------------------------------------------------------

class _SplitBuffer:
    """Owner of one ``fe_split_alloc`` array; torch reads it through ``__cuda_array_interface__`` and keeps this
    object alive for as long as any tensor (or view) of the array lives; the memory returns to the pool with it."""

    def __init__(self, nbytes: int, device_index: int) -> None:
        from feinsum_amd import _hip

        self.ptr, self.nbytes, self.device_index = _hip.split_alloc(nbytes), int(nbytes), int(device_index)
        self._free = _hip.split_free     # (bound now: module globals may be gone at interpreter exit)
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 3, "strides": None}

    def __del__(self) -> None:
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if not ptr:
            return
        try:
            import torch

            with torch.cuda.device(self.device_index):
                self._free(ptr)          # waits for the device like hipFree, then unmaps
        except Exception:                # noqa: BLE001  (interpreter shutdown: the process is going away anyway)
            pass


def empty(shape: Sequence[int], dtype: Any = None, device: Any = None, *, written: bool = True) -> Any:
    """
    A new uninitialised device tensor, as ``torch.empty`` -- for arrays a launch WRITES taken from the split allocator:
    the array's 4 MiB pieces alternate between two classes of physical memory, so every write stream of a launch is
    spread over both all the time.  That is what makes the DG launches run at 76-78 % of the HBM roofline instead of
    67-74 %: grad's three output planes, the four face-mass outputs and div's single output
    alike, with the default kernels.  No arena and no timing scan: the memory mapped is the array's size rounded up to
    2 MiB; arrays below 8 MiB, ``written=False`` and CPU devices get a plain ``torch.empty`` (where a read-only array
    lies does not matter).  The tensor is an ordinary torch tensor (views, copies, kernels); its memory returns to the
    allocator's pool when the last view is gone.
    """
    import torch

    dtype = dtype or torch.float64
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    shape = tuple(int(d) for d in shape)
    nbytes = int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size()
    if not written or dev.type != "cuda" or nbytes < SPLIT_MIN_BYTES:
        return torch.empty(shape, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        buf = _SplitBuffer(nbytes, dev.index)
        flat = torch.as_tensor(buf, device=dev)
    if flat.data_ptr() != buf.ptr:
        raise RuntimeError("torch copied the split allocator's array instead of wrapping it")
    return flat.view(dtype).view(shape)


def zeros(shape: Sequence[int], dtype: Any = None, device: Any = None, *, written: bool = True) -> Any:
    """:func:`empty`, zero-filled (what the reference's ``cla.zeros`` outputs are: ``src/feinsum/measure.py:44-60``)."""
    return empty(shape, dtype, device, written=written).zero_()


#: below two pieces of 4 MiB there is nothing to alternate (include/feinsum_hip.h)
SPLIT_MIN_BYTES = 8 * MIB


def split_info(tensor: Any) -> Dict[str, Any]:
    """What the allocator did for *tensor* (an array of :func:`empty`): ``{"bytes", "mapped_bytes", "piece_mib", "pieces",
    "pieces_by_class", "first_pieces": the classes of the first 16 pieces, "tail_bytes", "alloc_ms"}``; ``{}`` for any
    other tensor."""
    import torch

    from feinsum_amd import _hip
    from feinsum_amd.diagnostics import InvalidParameterError

    try:
        with torch.cuda.device(tensor.device):
            return _hip.split_info(tensor.untyped_storage().data_ptr())
    except (InvalidParameterError, RuntimeError):
        return {}


def split_stats(device: Any = None) -> Dict[str, Any]:
    """The pool of the split allocator on *device*: classes seen, free pieces, pieces created, probes, spacers, ms."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        return _hip.split_stats()


def split_reserve(nbytes: int, device: Any = None) -> None:
    """Announce the total size of the written arrays about to be allocated with :func:`empty` / :func:`zeros` on *device*
    (``fe_split_reserve``): the allocator collects half of it of each of two classes of physical memory at once, instead
    of array by array -- a class its search has left behind does not come back."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _hip.split_reserve(int(nbytes))


def split_trim(device: Any = None) -> None:
    """Release the allocator's free pieces on *device* to the driver."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _hip.split_trim()

ALIGN = 2 * MIB
#: candidate gaps between consecutive arrays, MiB: plateaus of the measured landscapes are >= 100 MiB wide, but where
#: they lie differs from process to process (it follows the physical pages behind the arena), so the search is dense
DEFAULT_GAPS_MIB = tuple(range(0, 2049, 64))


class Arena:
    """One device allocation; arrays are views into it."""

    def __init__(self, nbytes: int, device: Any) -> None:
        import torch

        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def view(self, offset: int, shape: Sequence[int], dtype: Any) -> Any:
        import torch

        n = int(torch.Size(tuple(shape)).numel()) * torch.empty((), dtype=dtype).element_size()
        return self.buf[offset:offset + n].view(dtype).view(tuple(shape))


def layout_offsets(sizes: Sequence[int], gap: int) -> List[int]:
    """Start of every array: 2 MiB aligned, `gap` bytes behind the end of its predecessor."""
    offs, off = [], 0
    for k, nb in enumerate(sizes):
        off = (off + ALIGN - 1) // ALIGN * ALIGN + (gap if k else 0)
        offs.append(off)
        off += int(nb)
    return offs


def arena_bytes(sizes: Sequence[int], max_gap: int) -> int:
    return layout_offsets(sizes, max_gap)[-1] + int(sizes[-1]) + ALIGN


def tune_gap(arrays: Sequence[Tuple[str, Tuple[int, ...], Any]], device: Any,
             make_step: Callable[[Dict[str, Any]], Callable[[int], float]], *,
             gaps_mib: Sequence[int] = DEFAULT_GAPS_MIB, fill: Callable[[str, Any], None] | None = None,
             rounds: int = 3, launches: int = 20, warmup: int = 10):
    """
    Pick the gap between consecutive arrays that makes the launch fastest.

    *arrays*: ``(name, shape, torch dtype)`` in layout order.  *make_step(views)* binds the launch to
    the views ``{name: tensor}`` and returns ``step(n) -> device seconds for n launches``.
    *fill(name, view)* initialises an array (inputs should hold realistic values: the clocks depend
    on the data).  Returns ``(arena, views, report)`` with the views laid out at the best gap and
    (re)filled; ``report`` lists the median milliseconds per launch of every candidate.
    """
    import torch

    sizes = [int(torch.Size(s).numel()) * torch.empty((), dtype=dt).element_size() for _, s, dt in arrays]
    arena = Arena(arena_bytes(sizes, max(gaps_mib) * MIB), device)

    def views_at(gap_mib: int) -> Dict[str, Any]:
        offs = layout_offsets(sizes, gap_mib * MIB)
        views = {name: arena.view(off, shape, dt) for (name, shape, dt), off in zip(arrays, offs)}
        if fill is not None:
            for name, v in views.items():
                fill(name, v)
        return views

    timings: Dict[int, float] = {}
    for gap in gaps_mib:
        step = make_step(views_at(gap))
        step(warmup)
        ts = sorted(step(launches) / launches for _ in range(rounds))
        timings[gap] = ts[len(ts) // 2] * 1e3
    best = min(timings, key=timings.get)
    views = views_at(best)
    report = {"mode": "tuned", "what": "all arrays in one arena, 2 MiB aligned, uniform gap between consecutive arrays; "
                                      "gap chosen by timing the launch (feinsum_amd/placement.py)",
              "best_gap_mib": best, "ms_by_gap_mib": {str(g): round(t, 5) for g, t in timings.items()}}
    return arena, views, report


def split_order(stages: Sequence[Sequence[Tuple[str, Tuple[int, ...], Any]]]) -> List[Tuple[str, Tuple[int, ...], Any]]:
    """
    Layout order of the WRITTEN arrays of a (multi-stage) launch such that ONE cut through the layout splits the
    concurrently written streams of every stage: of a stage that writes several arrays (face-mass x b) the first half
    goes to the left and the second half to the right; a stage that writes one array of several planes (grad:
    ``[3][E][Np]``) goes in the middle, where the cut can fall inside it; single-stream outputs (div) go to the left.
    *stages*: per stage the ``(name, shape, dtype)`` of its outputs.
    """
    left: List[Any] = []
    middle: List[Any] = []
    right: List[Any] = []
    for outs in stages:
        outs = list(outs)
        if len(outs) > 1:
            half = (len(outs) + 1) // 2
            left += outs[:half]
            right += outs[half:]
        elif outs and len(outs[0][1]) == 3 and outs[0][1][0] > 1:      # [planes][E][Np]: written plane by plane together
            middle += outs
        else:
            left += outs
    return left + middle + right


def tune_base(arrays: Sequence[Tuple[str, Tuple[int, ...], Any]], device: Any,
              make_step: Callable[[Dict[str, Any]], Callable[[int], float]], *,
              arena_gib: float = 66.0, gap_mib: int = 64, fill: Callable[[str, Any], None] | None = None,
              coarse_launches: int = 10, launches: int = 20, rounds: int = 3, stride_mib: int | None = None,
              fine_step_mib: int | None = None):
    """
    Pick the POSITION of the layout inside one large arena that makes the launch fastest.

    The arrays keep a fixed spacing (*gap_mib* between consecutive arrays); the layout as a whole is
    moved through an arena of *arena_gib* GiB (clamped to 60 % of the free device memory) in steps of
    half its own length, the launch is timed at every position (a short batch), and the neighbourhood
    of the best position is refined.  Arguments and return value as :func:`tune_gap`; ``report`` holds
    the coarse scan.
    """
    import torch

    sizes = [int(torch.Size(s).numel()) * torch.empty((), dtype=dt).element_size() for _, s, dt in arrays]
    rel = layout_offsets(sizes, gap_mib * MIB)
    length = rel[-1] + sizes[-1]
    free, _total = torch.cuda.mem_get_info(device)
    nbytes = int(min(arena_gib * (1 << 30), 0.6 * free))
    nbytes = max(nbytes, length + 2 * ALIGN)
    arena = Arena(nbytes, device)
    last = (nbytes - length - ALIGN) // ALIGN * ALIGN

    def views_at(base: int) -> Dict[str, Any]:
        views = {name: arena.view(base + off, shape, dt) for (name, shape, dt), off in zip(arrays, rel)}
        if fill is not None:
            for name, v in views.items():
                fill(name, v)
        return views

    def time_at(base: int, n: int, reps: int) -> float:
        step = make_step(views_at(base))
        step(5)
        ts = sorted(step(n) / n for _ in range(reps))
        return ts[len(ts) // 2] * 1e3

    # the launch is fast while the class boundary cuts through the written arrays (the plateau is as wide as one array /
    # one plane, with linear ramps of the same width either side): coarse steps of a quarter of the layout, at least
    # 256 MiB and at most 1 GiB, then the neighbourhood of the best position in steps of at most 128 MiB
    # (*stride_mib* / *fine_step_mib* override: a launch with ONE written array that walks it in two windows -- div,
    # FE_VARIANT_MFMA_SPLIT -- is fast only while the boundary lies near the middle of that array: a narrow peak)
    stride = min(max(length // 4 // ALIGN * ALIGN, 256 * MIB), 1024 * MIB)
    if stride_mib is not None:
        stride = max(ALIGN, stride_mib * MIB // ALIGN * ALIGN)
    coarse = {b: time_at(b, coarse_launches, 1) for b in range(0, last + 1, stride)}
    best = min(coarse, key=coarse.get)
    fine = {best: time_at(best, launches, rounds)}
    step = max(64 * MIB, min(stride // 4, 128 * MIB)) // ALIGN * ALIGN
    if fine_step_mib is not None:
        step = max(ALIGN, fine_step_mib * MIB // ALIGN * ALIGN)
    for k in range(-(stride // step), stride // step + 1):
        b = (best + k * step) // ALIGN * ALIGN
        if 0 <= b <= last and b not in fine:
            fine[b] = time_at(b, launches, rounds)
    best = min(fine, key=fine.get)
    views = views_at(best)
    ordered = sorted(coarse.values())
    report = {"mode": "tuned", "what": f"all arrays in one {nbytes / 2**30:.0f} GiB arena, {gap_mib} MiB apart; the layout "
                                      "is moved through the arena and kept where the launch times fastest "
                                      "(feinsum_amd/placement.py: across a joint of the allocator's physical blocks)",
              "best_base_mib": best // MIB, "best_ms": round(fine[best], 5),
              "scan_positions": len(coarse), "scan_median_ms": round(ordered[len(ordered) // 2], 5),
              "scan_min_ms": round(ordered[0], 5), "scan_max_ms": round(ordered[-1], 5),
              "fast_positions_mib": [b // MIB for b, t in coarse.items() if t < 0.97 * ordered[len(ordered) // 2]],
              # False: no position stood out (an arena of one class of physical memory, or a launch that writes a
              # single stream): the layout then simply sits where it timed best
              "class_boundary_found": bool(fine[best] < 0.97 * ordered[len(ordered) // 2])}
    return arena, views, report


def tune_base_retry(arrays: Sequence[Tuple[str, Tuple[int, ...], Any]], device: Any,
                    make_step: Callable[[Dict[str, Any]], Callable[[int], float]], *, attempts: int = 3, **kwargs: Any):
    """
    :func:`tune_base`, again in a fresh arena while no class boundary was found (an arena can lie in ONE class of
    physical memory -- runs of up to 72 GiB were seen -- and then no position in it is fast).  The arenas tried so far
    stay allocated meanwhile, so that the next one comes from other physical memory; the fastest result is kept and
    the other arenas are released.  ``report["arenas_tried"]`` says how many it took.
    """
    import torch

    kept = None
    held = []
    for k in range(max(1, attempts)):
        try:
            arena, views, report = tune_base(arrays, device, make_step, **kwargs)
        except torch.cuda.OutOfMemoryError:
            if kept is None:
                raise
            break
        if kept is None or report["best_ms"] < kept[2]["best_ms"]:
            if kept is not None:
                held.append(kept[0])
            kept = (arena, views, report)
        else:
            held.append(arena)
        if report["class_boundary_found"]:
            break
    kept[2]["arenas_tried"] = len(held) + 1
    del held        # back to the caller's caching allocator (not emptied here: the cache is the caller's)
    return kept
