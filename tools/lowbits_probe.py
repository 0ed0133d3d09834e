#!/usr/bin/env python
"""
Do the LOW address bits of the written arrays matter?  Face-mass x 4 with its four outputs inside ONE physical block
(deep inside the arena's first block), output k shifted by k * delta bytes for delta = 0, 128 B ... 1 MiB; then the
same for grad through the planes launcher (plane k at k * (E Np 8 rounded to 2 MiB) + k * delta).

    python tools/lowbits_probe.py
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
    from feinsum_amd import _hip, measure

    E, Np = 1_000_000, 35
    expr = dg.face_mass(4)
    q = f.DeviceQueue(0)
    names = sorted(expr.all_args)
    shape_of = {n: tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[n]) for n in names}
    out_shape = (E, Np)
    onames = list(expr.output_names)
    nbytes = lambda s: 8 * int(torch.Size(s).numel())   # noqa: E731
    arena = torch.empty(24 * GIB, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    al = lambda x: int(x) // (2 * MIB) * (2 * MIB)   # noqa: E731
    W = al(nbytes(out_shape) + 2 * MIB - 1) + 2 * MIB
    view = lambda off, shape: arena[off:off + nbytes(shape)].view(torch.float64).view(shape)   # noqa: E731
    deltas = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384, 20480, 32768, 65536, 69632, 131072, 1 << 20, (1 << 20) + 4096]

    def timed(bound):
        for _ in range(10):
            bound.launch(q.stream_ptr)
        q.finish()
        ts = sorted(bound.time_batch(20, q.stream_ptr) / 20 * 1e3 for _ in range(3))
        return ts[1]

    def fm(delta_out, delta_in):
        off, dev = 0, {}
        k = 0
        for n in names:
            off = al(off + 2 * MIB - 1) + 64 * MIB
            shift = (k * delta_in) if n.startswith("v") else 0
            k += n.startswith("v")
            dev[n] = view(off + shift, shape_of[n])
            dev[n].uniform_(0.0, 1.0, generator=g)
            off += nbytes(shape_of[n]) + 2 * MIB
        base = 8 * GIB
        outs = {name: view(base + i * W + i * delta_out, out_shape) for i, name in enumerate(onames)}
        return timed(measure._bind(expr, q, dev, outs, None)[1])

    print("# face-mass x 4, outputs in one block; output k shifted by k * delta (inputs v_k: k * delta_in)")
    for d in deltas:
        print(f"delta_out {d:8d} B  delta_in 0: {fm(d, 0):.4f} ms     delta_in = delta_out: {fm(d, d):.4f} ms", flush=True)
    print(f"delta_out 0  delta_in 4096: {fm(0, 4096):.4f} ms")

    # grad through the planes launcher
    lib = _hip.load_library()
    J = view(64 * MIB, (3, 3, E)); J.uniform_(0.0, 1.0, generator=g)
    D = view(256 * MIB, (3, Np, Np)); D.uniform_(0.0, 1.0, generator=g)
    u = view(320 * MIB, (E, Np)); u.uniform_(0.0, 1.0, generator=g)
    j3 = _hip._ptr_array([J.data_ptr() + 8 * 3 * E * x for x in range(3)])
    up = _hip._ptr_array([u.data_ptr()])
    base = arena.data_ptr() + 8 * GIB
    P = E * Np * 8

    def grad_planes(stride):
        outs = _hip._ptr_array([base + k * stride for k in range(3)])
        fn = lambda: _hip.check(lib.fe_gradplanes3d_f64(j3, D.data_ptr(), up, outs, E, Np, 1, 0, 0, 0))   # noqa: E731
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                fn()
            t1.record()
            t1.synchronize()
            ts.append(t0.elapsed_time(t1) / 20)
        return sorted(ts)[1]

    print("# grad, three output planes in one block, plane stride = S")
    print(f"S = E Np 8 = {P} (the array [3][E][Np]; S mod 64 KiB = {P % 65536}): {grad_planes(P):.4f} ms", flush=True)
    S0 = al(P + 2 * MIB - 1) + 2 * MIB
    for d in deltas:
        print(f"S = {S0 // MIB} MiB + {d:8d} B: {grad_planes(S0 + d):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
