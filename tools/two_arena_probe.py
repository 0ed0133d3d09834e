#!/usr/bin/env python
"""
Arrays split over SEVERAL separately allocated arenas (different physical extents by construction?):

    python tools/two_arena_probe.py <facemass|grad|div> <arena GiB> <n arenas> [E]

Assignments tried: "one" (everything in arena 0), "alternate" (array k -> arena k mod n), "rw" (inputs in arena 0,
outputs round-robin over the others), "blocks" (first half / second half).  Arrays inside an arena are 64 MiB apart.
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

    fam, gib, n_ar = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    E = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    expr = {"facemass": dg.face_mass(4), "grad": dg.grad(), "div": dg.div()}[fam]
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)
    shape_of = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]) for n in names}
    out_shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    arenas = [torch.empty(gib << 30, dtype=torch.uint8, device="cuda") for _ in range(n_ar)]
    print("arenas at", [hex(a.data_ptr()) for a in arenas], flush=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    arrays = [(n, shape_of[n], False) for n in names] + [(n, out_shape, True) for n in expr.output_names]

    def run(assign):
        offs = [0] * n_ar
        dev, outs = {}, {}
        for k, (n, shape, is_out) in enumerate(arrays):
            a = assign(k, is_out)
            nb = 8 * int(torch.Size(shape).numel())
            off = (offs[a] + 2 * MIB - 1) // (2 * MIB) * (2 * MIB)
            t = arenas[a][off:off + nb].view(torch.float64).view(shape)
            offs[a] = off + nb + 64 * MIB
            if is_out:
                outs[n] = t
            else:
                t.uniform_(0.0, 1.0, generator=g)
                dev[n] = t
        _, bound, _ = measure._bind(expr, q, dev, outs, None)
        for _ in range(20):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(20, q.stream_ptr) / 20 * 1e3 for _ in range(5))
        return ts[2]

    n_in = len(names)
    wk = [0]
    def rw(k, is_out):
        if not is_out:
            return 0
        wk[0] += 1
        return 1 + (wk[0] - 1) % max(n_ar - 1, 1) if n_ar > 1 else 0
    for rep in range(2):
        wk[0] = 0
        print(f"{fam} {gib} GiB x {n_ar}: one {run(lambda k, o: 0):.4f}  alternate {run(lambda k, o: k % n_ar):.4f}  "
              f"rw {run(rw):.4f}  blocks {run(lambda k, o: min(k * n_ar // len(arrays), n_ar - 1)):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
