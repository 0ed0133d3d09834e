"""
Where the operand arrays of a launch sit in device memory.

New functionality (the reference allocates every array separately through PyOpenCL,
``src/feinsum/measure.py:44-60,80-108``, and has no notion of placement).  On MI355X the same
launch on the same device runs up to 14 % apart depending on where its arrays lie relative to one
another (``profiles/r02/placement_*.txt``: face-mass x 4 at E = 1e6 0.485 ... 0.569 ms, grad 0.191
... 0.215 ms, with the spacing between consecutive arrays as the only variable): a DG launch streams
13 (grad) to 26 (face-mass) arrays and array slabs at once, and how those streams fall onto the
memory channels and DRAM banks follows from their physical addresses.  Separate allocations land
wherever the allocator puts them, which is what made the round-1 numbers differ "between devices".

:class:`Arena` carves all arrays of a workload out of ONE allocation, 2 MiB aligned, with a
uniform gap between consecutive arrays; :func:`tune_gap` times the bound launch for a few candidate
gaps and keeps the fastest -- an autotuning step in the spirit of the reference's transform search
(``src/feinsum/tuning/__init__.py:573-633``), over memory layout instead of loop structure.
The kernels and their results do not depend on placement.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, List, Sequence, Tuple

MIB = 1 << 20
ALIGN = 2 * MIB
#: candidate gaps between consecutive arrays, MiB (the landscape is made of plateaus >= 100 MiB wide)
DEFAULT_GAPS_MIB = (0, 136, 296, 456, 616, 776, 936, 1096, 1176, 1256)


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
