#!/usr/bin/env python
"""
Does the placement of the operand buffers in device memory change the kernel time?

    python tools/placement_probe.py [facemass|grad|div] [E] [trials]

The same face-mass launch measured 0.509 ms in tools/fe_check (hipMalloc'd buffers) and 0.562 ms in
bench.py (torch allocations) on one device in one session.  This probe times the launch over several
allocations of the same operands -- each preceded by a dummy allocation of a different size, so the
buffers land at different offsets -- and once more with every operand carved out of ONE contiguous
arena at 2 MiB-multiple offsets.  Prints the addresses (mod 1 GiB, in MiB) next to the time.
"""

from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main() -> None:
    import torch

    import dg
    import feinsum_amd as f
    from feinsum_amd import measure

    fam = sys.argv[1] if len(sys.argv) > 1 else "facemass"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    expr = {"facemass": dg.face_mass(4), "grad": dg.grad(), "div": dg.div()}[fam]
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)

    def shapes():
        for name in names:
            yield name, tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[name])

    def time_it(dev, outs):
        _, bound, _ = measure._bind(expr, q, dev, outs, None)
        for _ in range(30):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(20, q.stream_ptr) / 20 * 1e3 for _ in range(7))
        return ts[len(ts) // 2], ts[0]

    out_shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
    g = torch.Generator(device="cuda").manual_seed(0)
    mib = lambda t: (t.data_ptr() % (1 << 30)) / (1 << 20)   # noqa: E731
    for trial in range(trials):
        torch.cuda.empty_cache()
        dummy = torch.empty((trial * 37 + 1) * (1 << 20) + trial * 4096 * 13, dtype=torch.uint8, device="cuda")
        dev = {n: torch.rand(s, dtype=torch.float64, device="cuda", generator=g) for n, s in shapes()}
        outs = {n: torch.zeros(out_shape, dtype=torch.float64, device="cuda") for n in expr.output_names}
        med, mn = time_it(dev, outs)
        where = " ".join(f"{n}@{mib(t):.2f}" for n, t in list(dev.items()) + list(outs.items()))
        print(f"trial {trial}: dummy {dummy.numel() / 2**20:8.2f} MiB  median {med:.4f} ms  min {mn:.4f} ms   {where}", flush=True)
        del dev, outs, dummy
    # one arena, operands at 2 MiB-multiple offsets, in different orders
    torch.cuda.empty_cache()
    sizes = {n: 8 * int(torch.Size(s).numel()) for n, s in shapes()}
    osize = 8 * int(torch.Size(out_shape).numel())
    total = sum(sizes.values()) + len(expr.output_names) * osize + (len(sizes) + 8) * (4 << 20)
    arena = torch.empty(total, dtype=torch.uint8, device="cuda")
    for order in ("inputs_then_outputs", "interleaved", "skewed_4k"):
        off, dev, outs = 0, {}, {}

        def carve(nbytes, shape, skew=0):
            nonlocal off
            off = (off + (2 << 20) - 1) // (2 << 20) * (2 << 20) + skew
            t = arena[off:off + nbytes].view(torch.float64).view(shape)
            off += nbytes
            return t

        k = 0
        onames = list(expr.output_names)
        for n, s in shapes():
            dev[n] = carve(sizes[n], s, skew=(4096 * 3 * k if order == "skewed_4k" else 0))
            dev[n].copy_(torch.rand(s, dtype=torch.float64, device="cuda", generator=g))
            k += 1
            if order == "interleaved" and onames and k >= 3:
                on = onames.pop(0)
                outs[on] = carve(osize, out_shape)
        for on in onames:
            outs[on] = carve(osize, out_shape, skew=(4096 * 5 * k if order == "skewed_4k" else 0))
            k += 1
        med, mn = time_it(dev, outs)
        where = " ".join(f"{n}@{mib(t):.2f}" for n, t in list(dev.items()) + list(outs.items()))
        print(f"arena {order}: median {med:.4f} ms  min {mn:.4f} ms   {where}", flush=True)


if __name__ == "__main__":
    main()
