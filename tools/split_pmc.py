#!/usr/bin/env python
"""
The program under rocprofv3 for tools/split_pmc.sh: face-mass x 4 with its four outputs (a) all below a joint of the
allocator's physical blocks, (b) two below + two above, (c) all above -- 40 launches each, as the LAST 120 dispatches of
the process (the joint is found first with a short scan, see tools/split_probe.py).
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

    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 70
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
    W = al(nbytes(out_shape) + 2 * MIB - 1)
    view = lambda off, shape: arena[off:off + nbytes(shape)].view(torch.float64).view(shape)   # noqa: E731
    off, dev = 0, {}
    for n in names:
        off = al(off + 2 * MIB - 1)
        dev[n] = view(off, shape_of[n])
        dev[n].uniform_(0.0, 1.0, generator=g)
        off += nbytes(shape_of[n]) + 64 * MIB
    in_end = off

    def bound_at(offsets):
        outs = {name: view(al(o), out_shape) for name, o in zip(onames, offsets)}
        return measure._bind(expr, q, dev, outs, None)[1]

    # the joint is located with the GRAD launch (another kernel name: the profiler is told to look at face-mass only,
    # so the scan runs at full speed): its [3][E][35] output is fast while the joint cuts the middle plane
    gexpr = dg.grad()
    gshape = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in gexpr.arg_to_shape[n]) for n in sorted(gexpr.all_args)}
    gdev = {}
    for n, shp in gshape.items():
        off = al(off + 2 * MIB - 1)
        gdev[n] = view(off, shp)
        gdev[n].uniform_(0.0, 1.0, generator=g)
        off += nbytes(shp) + 64 * MIB
    in_end = off
    gout_shape = (3, E, 35)

    def time_grad(start, n=3):
        b = measure._bind(gexpr, q, gdev, {gexpr.output_names[0]: view(al(start), gout_shape)}, None)[1]
        b.launch(q.stream_ptr)
        q.finish()
        return b.time_batch(n, q.stream_ptr) / n * 1e3

    first, last = al(in_end + 4 * W), gib * GIB - 5 * W - 64 * MIB
    coarse = {s: time_grad(s) for s in range(first, last, 256 * MIB)}
    med = sorted(coarse.values())[len(coarse) // 2]
    best = min(coarse, key=coarse.get)
    if coarse[best] > 0.95 * med:
        print("no joint found", flush=True)
        return
    fine = {s: time_grad(s, n=5) for s in range(max(first, best - 768 * MIB), min(last, best + 768 * MIB), 32 * MIB)}
    plateau = [s for s, t in fine.items() if t < min(fine.values()) + 0.25 * (med - min(fine.values()))]
    jt = al((min(plateau) + max(plateau)) // 2 + nbytes(gout_shape) // 2)
    print(f"joint at {jt / GIB:.3f} GiB (grad scan median {med:.4f}, best {coarse[best]:.4f} ms)", flush=True)
    configs = {"4 below": [jt - 4 * W, jt - 3 * W, jt - 2 * W, jt - W], "2 + 2": [jt - 2 * W, jt - W, jt, jt + W],
               "4 above": [jt, jt + W, jt + 2 * W, jt + 3 * W]}
    q.finish()
    for what, offs in configs.items():
        b = bound_at(offs)
        t = b.time_batch(40, q.stream_ptr) / 40 * 1e3
        print(f"{what}: {t:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
