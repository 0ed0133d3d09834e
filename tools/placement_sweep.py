#!/usr/bin/env python
"""
Kernel time against the relative placement of the operand arrays inside ONE contiguous arena.

    python tools/placement_sweep.py <facemass|grad|div> <mode> [E] [wide]

mode "gap":   every array starts `gap` bytes behind the (2 MiB-rounded) end of the previous one,
              gap = 0, 0.25, 0.5, ... MiB (fine) then 2 MiB steps (coarse)
mode "out":   inputs packed, only the outputs are shifted by the gap
mode "base":  everything packed, the whole layout shifted inside the arena by the gap
mode "region": arrays 64 MiB apart, the whole layout moved through a 48 GiB arena in 1 GiB steps
One line per point: gap, median ms, min ms.
"""

from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
MIB = 1 << 20


def main() -> None:
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    fam, mode = sys.argv[1], sys.argv[2]
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    expr = {"facemass": dg.face_mass(4), "grad": dg.grad(), "div": dg.div()}[fam]
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)
    shape_of = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]) for n in names}
    out_shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    nbytes = lambda s: 8 * int(torch.Size(s).numel())   # noqa: E731
    n_arrays = len(names) + len(expr.output_names)
    wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
    import os
    max_gap = (1300 if wide else 72) * MIB
    if os.environ.get("FE_GAPS"):
        max_gap = max(int(x) for x in os.environ["FE_GAPS"].split(",")) * MIB
    if mode == "region":
        max_gap = 64 * MIB
    total = sum(nbytes(s) for s in shape_of.values()) + len(expr.output_names) * nbytes(out_shape) \
        + n_arrays * (max_gap + 4 * MIB) + 8 * MIB + ((int(os.environ.get("FE_REGION_GIB", "48")) + 1) * 1024 * MIB if mode == "region" else 0)
    arena = torch.empty(total, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    src = {n: torch.rand(s, dtype=torch.float64, device="cuda", generator=g) for n, s in shape_of.items()}

    def layout(gap):
        off, dev, outs = (gap if mode in ("base", "region") else 0), {}, {}

        def carve(nb, shape, extra):
            nonlocal off
            off = (off + 2 * MIB - 1) // (2 * MIB) * (2 * MIB) + extra
            t = arena[off:off + nb].view(torch.float64).view(shape)
            off += nb
            return t

        for k, n in enumerate(names):
            dev[n] = carve(nbytes(shape_of[n]), shape_of[n],
                           gap if (mode == "gap" and k > 0) else (64 * MIB if (mode == "region" and k > 0) else 0))
            dev[n].copy_(src[n])
        for on in expr.output_names:
            outs[on] = carve(nbytes(out_shape), out_shape,
                             gap if mode in ("gap", "out") else (64 * MIB if mode == "region" else 0))
        return dev, outs

    def time_it(dev, outs):
        _, bound, _ = measure._bind(expr, q, dev, outs, None)
        for _ in range(20):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(20, q.stream_ptr) / 20 * 1e3 for _ in range(5))
        return ts[len(ts) // 2], ts[0]

    gaps = [int(x * MIB / 4) for x in range(0, 17)] + [x * MIB for x in range(6, 72, 2)]
    if wide:
        gaps = [x * MIB for x in range(0, 1300, 8)]
    if mode == "region":
        gaps = [x * 1024 * MIB for x in range(0, int(os.environ.get("FE_REGION_GIB", "48")) + 1, int(os.environ.get("FE_REGION_STEP", "1")))] * 2
    import os
    if os.environ.get("FE_GAPS"):          # explicit list, MiB
        gaps = [int(x) * MIB for x in os.environ["FE_GAPS"].split(",")]
    for gap in gaps:
        dev, outs = layout(gap)
        med, mn = time_it(dev, outs)
        print(f"{fam} {mode} gap {gap / MIB:7.2f} MiB  median {med:.4f} ms  min {mn:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
