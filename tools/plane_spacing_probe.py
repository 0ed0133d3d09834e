#!/usr/bin/env python
"""
Kernel time against the spacing of the WRITE streams only (read streams packed 64 MiB apart):

    python tools/plane_spacing_probe.py <grad|facemass> [E]

grad: the three output planes out[x] (one array [3][E][Np] when the spacing is E Np 8 bytes) through the planes
launcher fe_gradplanes3d_f64; facemass: the four output arrays.  Spacing from "contiguous" to 6.5 GiB.
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
MIB = 1 << 20


def main() -> None:
    import torch

    from feinsum_amd import _hip

    fam = sys.argv[1] if len(sys.argv) > 1 else "grad"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    Np, nf, Nfp, nb = 35, 4, 15, 4
    n_w = 3 if fam == "grad" else nb
    w_bytes = E * Np * 8
    max_s = 6656 * MIB
    arena = torch.empty(n_w * max_s + w_bytes + 8 * 1024 * MIB, dtype=torch.uint8, device="cuda")
    base = arena.data_ptr()
    g = torch.Generator(device="cuda").manual_seed(0)
    lib = _hip.load_library()
    # reads: packed 64 MiB apart behind the write region
    roff = n_w * max_s + w_bytes + 64 * MIB

    def carve_read(shape):
        nonlocal roff
        n = 8 * int(torch.Size(shape).numel())
        roff = (roff + 2 * MIB - 1) // (2 * MIB) * (2 * MIB)
        t = arena[roff:roff + n].view(torch.float64).view(shape)
        t.uniform_(0.0, 1.0, generator=g)
        roff += n + 64 * MIB
        return t

    if fam == "grad":
        J, D, u = carve_read((3, 3, E)), carve_read((3, Np, Np)), carve_read((E, Np))
        j3 = _hip._ptr_array([J.data_ptr() + 8 * 3 * E * x for x in range(3)])
        up = _hip._ptr_array([u.data_ptr()])
    else:
        Jf, R = carve_read((E, nf)), carve_read((nf, Np, Nfp))
        v = [carve_read((nf, E, Nfp)) for _ in range(nb)]
        vp = _hip._ptr_array([t.data_ptr() for t in v])

    def launch(S):
        outs = _hip._ptr_array([base + k * S for k in range(n_w)])
        if fam == "grad":
            return lambda: _hip.check(lib.fe_gradplanes3d_f64(j3, D.data_ptr(), up, outs, E, Np, 1, 0, 0, 0))
        return lambda: _hip.check(lib.fe_facemass_f64(Jf.data_ptr(), R.data_ptr(), vp, outs, E, Np, nf, Nfp, nb, 0, 0, 0))

    def time_it(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                fn()
            t1.record()
            t1.synchronize()
            ts.append(t0.elapsed_time(t1) / 20)
        ts.sort()
        return ts[2]

    spacings = [w_bytes] + [s * MIB for s in range(512, 6657, 256)]
    if len(sys.argv) > 3 and sys.argv[3] == "phase":     # sub-8-MiB phases of the spacing
        spacings = [w_bytes] + [int((512 + d) * MIB) for d in (0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5, 8,
                                                             10, 12, 14, 16, 20, 24, 28, 32, 36, 44, 52, 60, 68)]
    for rep in range(2):
        for S in spacings:
            ms = time_it(launch(S))
            print(f"{fam} rep {rep} write-stream spacing {S / MIB:8.1f} MiB  {ms:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
