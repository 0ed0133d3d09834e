#!/usr/bin/env python
"""
Is it the SPLIT of the written arrays between two physical blocks that makes a launch fast, or their nearness to
the joint?  Face-mass x 4 (four separately placeable outputs of 267 MiB), everything in one arena:

    python tools/split_probe.py [arena GiB]

1. the four outputs, packed, are moved through the arena in 512 MiB steps (inputs fixed at the arena's start): the
   fast positions are the joints; every joint is then located to 32 MiB (centre of the plateau of a 64 MiB scan);
2. with the joints known, the four outputs are placed individually: k of them below a joint and 4 - k above, next
   to it or GiBs away from it, in different blocks altogether, interleaved ...
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

    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    E = 1_000_000
    expr = dg.face_mass(4)
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)
    shape_of = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]) for n in names}
    out_shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    onames = list(expr.output_names)
    nbytes = lambda s: 8 * int(torch.Size(s).numel())   # noqa: E731
    arena = torch.empty(gib * GIB, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    al = lambda x: int(x) // (2 * MIB) * (2 * MIB)   # noqa: E731
    W = al(nbytes(out_shape) + 2 * MIB - 1)             # one output, rounded up to 2 MiB

    def view(off, shape):
        return arena[off:off + nbytes(shape)].view(torch.float64).view(shape)

    # inputs: packed 64 MiB apart wherever `in_start` says
    def inputs_at(start):
        off, dev = start, {}
        for n in names:
            off = al(off + 2 * MIB - 1)
            dev[n] = view(off, shape_of[n])
            dev[n].uniform_(0.0, 1.0, generator=g)
            off += nbytes(shape_of[n]) + 64 * MIB
        return dev, off

    dev, in_end = inputs_at(0)

    def time_outs(offsets, n=20, reps=3):
        outs = {name: view(al(o), out_shape) for name, o in zip(onames, offsets)}
        _, bound, _ = measure._bind(expr, q, dev, outs, None)
        for _ in range(10):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(n, q.stream_ptr) / n * 1e3 for _ in range(reps))
        return ts[len(ts) // 2]

    packed = lambda s: [s + k * W for k in range(4)]   # noqa: E731
    first = al(in_end + GIB)
    last = gib * GIB - 4 * W - 64 * MIB
    print(f"# arena {gib} GiB at {hex(arena.data_ptr())}; inputs in [0, {in_end / GIB:.2f}) GiB; one output = {W / MIB:.0f} MiB", flush=True)
    coarse = {s: time_outs(packed(s), n=10, reps=1) for s in range(first, last, 512 * MIB)}
    med = sorted(coarse.values())[len(coarse) // 2]
    fast = sorted(s for s, t in coarse.items() if t < 0.95 * med)
    print(f"coarse scan of the packed outputs: median {med:.4f} ms, min {min(coarse.values()):.4f}; fast starts (GiB): "
          + " ".join(f"{s / GIB:.1f}:{coarse[s]:.3f}" for s in fast), flush=True)
    # cluster the fast starts (neighbours 512 MiB apart belong to one joint), then refine each
    clusters = []
    for s in fast:
        if clusters and s - clusters[-1][-1] <= 1024 * MIB:
            clusters[-1].append(s)
        else:
            clusters.append([s])
    joints = []
    for cl in clusters:
        c = (cl[0] + cl[-1]) // 2
        fine = {s: time_outs(packed(s), n=10, reps=1) for s in range(max(first, c - 1536 * MIB), min(last, c + 1536 * MIB), 64 * MIB)}
        lo_t = min(fine.values())
        plateau = [s for s, t in fine.items() if t < 0.5 * (med + lo_t)]
        jt = (min(plateau) + max(plateau)) // 2 + 2 * W          # the joint sits in the middle of the four outputs there
        joints.append(al(jt))
        print(f"joint at {jt / GIB:.3f} GiB (plateau {min(plateau) / MIB:.0f} .. {max(plateau) / MIB:.0f} MiB, {lo_t:.4f} ms)", flush=True)
    if not joints:
        return

    def show(what, offsets):
        ok = all(in_end <= o and o + W <= gib * GIB for o in offsets)
        print(f"{what:<78s} " + (f"{time_outs(offsets):.4f} ms" if ok else "(does not fit)"), flush=True)

    for jt in joints[:2]:
        print(f"== joint {jt / GIB:.3f} GiB", flush=True)
        below = lambda k, d=0: jt - d - (k + 1) * W        # k-th output slot below the joint, d bytes away from it   # noqa: E731
        above = lambda k, d=0: jt + d + k * W                                                                    # noqa: E731
        show("packed across it: 2 below + 2 above, adjacent", [below(1), below(0), above(0), above(1)])
        show("1 below + 3 above, adjacent", [below(0), above(0), above(1), above(2)])
        show("3 below + 1 above, adjacent", [below(2), below(1), below(0), above(0)])
        show("4 below, ending at the joint", [below(3), below(2), below(1), below(0)])
        show("4 above, starting at the joint", [above(0), above(1), above(2), above(3)])
        for d in (GIB, 4 * GIB, 8 * GIB):
            show(f"2 + 2, each pair {d // GIB} GiB away from the joint", [below(1, d), below(0, d), above(0, d), above(1, d)])
            show(f"1 + 3, {d // GIB} GiB away", [below(0, d), above(0, d), above(1, d), above(2, d)])
            show(f"2 below {d // GIB} GiB away + 2 above adjacent", [below(1, d), below(0, d), above(0), above(1)])
            show(f"2 below adjacent + 2 above {d // GIB} GiB away", [below(1), below(0), above(0, d), above(1, d)])
            show(f"4 below, {d // GIB} GiB away", [below(3, d), below(2, d), below(1, d), below(0, d)])
            show(f"4 above, {d // GIB} GiB away", [above(0, d), above(1, d), above(2, d), above(3, d)])
        show("interleaved: out0 below, out1 above, out2 below, out3 above (adjacent)", [below(1), above(0), below(0), above(1)])
        show("interleaved, 4 GiB away", [below(1, 4 * GIB), above(0, 4 * GIB), below(0, 4 * GIB), above(1, 4 * GIB)])
        show("spread inside the lower block: 2 GiB apart, ending 1 GiB below", [jt - GIB - W - k * 2 * GIB for k in range(4)])
    if len(joints) >= 2:
        a, b = joints[0], joints[1]
        print(f"== two joints {a / GIB:.3f} and {b / GIB:.3f} GiB (three blocks: I below a, II between, III above b)", flush=True)
        show("2 in I + 2 in III (nothing in II), 1 GiB from the joints", [a - GIB - 2 * W, a - GIB - W, b + GIB, b + GIB + W])
        show("2 in I + 2 in II (middle of II)", [a - GIB - 2 * W, a - GIB - W, (a + b) // 2, (a + b) // 2 + W])
        show("1 in I + 2 in II + 1 in III", [a - GIB - W, (a + b) // 2, (a + b) // 2 + W, b + GIB])
        show("4 in II (middle)", [(a + b) // 2 + k * W for k in range(4)])
    if len(joints) >= 3:
        a, b, c = joints[:3]
        show("one output in each of four blocks", [a - GIB - W, (a + b) // 2, (b + c) // 2, c + GIB])


if __name__ == "__main__":
    main()
