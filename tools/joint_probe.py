#!/usr/bin/env python
"""
What exactly is fast about a layout that lies across a joint of the allocator's physical blocks?

    python tools/joint_probe.py [facemass|grad] [arena GiB]

1. coarse scan (1 GiB steps) of the packed layout through the arena -> the fastest base;
2. fine scan (64 MiB steps) around it -> the plateau, i.e. where the joint lies relative to the arrays;
3. with the joint estimated as (plateau centre + end of the inputs): inputs and outputs placed independently --
   d bytes below / above the joint, both below, both above, swapped.
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
MIB, GIB = 1 << 20, 1 << 30


def main() -> None:
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    fam = sys.argv[1] if len(sys.argv) > 1 else "facemass"
    gib = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    E = 1_000_000
    expr = {"facemass": dg.face_mass(4), "grad": dg.grad(), "div": dg.div()}[fam]
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)
    shape_of = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]) for n in names}
    out_shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    nbytes = lambda s: 8 * int(torch.Size(s).numel())   # noqa: E731
    arena = torch.empty(gib * GIB, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    gap = 64 * MIB

    def pack(items, start):
        """[(name, shape)] packed from `start` (2 MiB aligned, 64 MiB apart) -> {name: view}, end"""
        off, views = start, {}
        for n, shape in items:
            off = (off + 2 * MIB - 1) // (2 * MIB) * (2 * MIB)
            views[n] = arena[off:off + nbytes(shape)].view(torch.float64).view(shape)
            off += nbytes(shape) + gap
        return views, off - gap

    ins = [(n, shape_of[n]) for n in names]
    outs_l = [(n, out_shape) for n in expr.output_names]
    in_len = pack(ins, 0)[1]
    out_len = pack(outs_l, 0)[1]

    def time_at(in_start, out_start, n=20, reps=3):
        dev, _ = pack(ins, in_start)
        outs, _ = pack(outs_l, out_start)
        for t in dev.values():
            t.uniform_(0.0, 1.0, generator=g)
        _, bound, _ = measure._bind(expr, q, dev, outs, None)
        for _ in range(10):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(n, q.stream_ptr) / n * 1e3 for _ in range(reps))
        return ts[len(ts) // 2]

    packed = lambda base, **kw: time_at(base, (base + in_len + gap + 2 * MIB - 1) // (2 * MIB) * (2 * MIB), **kw)   # noqa: E731
    total = in_len + gap + out_len
    print(f"# {fam}: inputs {in_len / MIB:.0f} MiB, outputs {out_len / MIB:.0f} MiB, arena {gib} GiB at {hex(arena.data_ptr())}", flush=True)
    coarse = {}
    for b in range(0, gib * GIB - total - GIB, GIB):
        coarse[b] = packed(b, n=10, reps=1)
    med = sorted(coarse.values())[len(coarse) // 2]
    fast = [b for b, t in coarse.items() if t < 0.96 * med]
    print("coarse scan: median %.4f ms; fast bases (GiB): %s" % (med, [(b // GIB, round(coarse[b], 4)) for b in fast]), flush=True)
    if not fast:
        return
    best = min(fast, key=coarse.get)
    fine = {}
    for b in range(max(0, best - 2 * GIB), best + 2 * GIB, 64 * MIB):
        fine[b] = packed(b, n=10, reps=1)
    print("fine scan (base MiB: ms): " + " ".join(f"{b // MIB}:{t:.3f}" for b, t in fine.items()), flush=True)
    plateau = [b for b, t in fine.items() if t < 0.5 * (med + min(fine.values()))]
    lo, hi = min(plateau), max(plateau)
    joint = (lo + hi) // 2 + in_len + gap // 2
    print(f"plateau: base {lo // MIB} .. {hi // MIB} MiB -> joint estimated at {joint / GIB:.3f} GiB "
          f"(ends of the inputs at the plateau edges: {(lo + in_len) / GIB:.3f} .. {(hi + in_len) / GIB:.3f} GiB)", flush=True)
    al = lambda x: int(x) // (2 * MIB) * (2 * MIB)   # noqa: E731
    print(f"straddling (inputs end at the joint, outputs start behind it): {time_at(al(joint - in_len - 32 * MIB), al(joint + 32 * MIB)):.4f} ms")
    for d in (256 * MIB, GIB, 4 * GIB, 12 * GIB):
        if joint - in_len - d < 0 or joint + d + out_len > gib * GIB:
            continue
        print(f"inputs end {d / GIB:5.2f} GiB below the joint, outputs start {d / GIB:5.2f} GiB above: "
              f"{time_at(al(joint - in_len - d), al(joint + d)):.4f} ms", flush=True)
        print(f"inputs end at the joint, outputs start {d / GIB:5.2f} GiB above: {time_at(al(joint - in_len - 32 * MIB), al(joint + d)):.4f} ms", flush=True)
        print(f"inputs end {d / GIB:5.2f} GiB below, outputs start at the joint: {time_at(al(joint - in_len - d), al(joint + 32 * MIB)):.4f} ms", flush=True)
    if joint - in_len - out_len - 6 * GIB > 0:
        print(f"both below the joint (outputs 2 GiB below it, inputs below them): "
              f"{time_at(al(joint - 2 * GIB - out_len - GIB - in_len), al(joint - 2 * GIB - out_len)):.4f} ms")
    if joint + 6 * GIB + total < gib * GIB:
        print(f"both above the joint: {time_at(al(joint + 2 * GIB), al(joint + 2 * GIB + in_len + GIB)):.4f} ms")
        print(f"swapped (outputs end below the joint, inputs start above): "
              f"{time_at(al(joint + 32 * MIB), al(joint - out_len - 32 * MIB)):.4f} ms")


if __name__ == "__main__":
    main()
