"""
Generates the golden fixtures under tests/golden/ (committed, small).

The reference (kaushikcfd/feinsum) ships no stored vectors and cannot be
imported in this environment (Python 3.12-only syntax + loopy / pyopencl /
opt_einsum absent: SURVEY §8c), so the vectors are produced with the
reference's own ground-truth expression,

    np.einsum(einsum.get_subscripts(), *inputs, optimize="optimal")

(reference: src/feinsum/measure.py:149-159), and every output is cross-checked
here against an independent extended-precision sum of products
(oracle.np_oracle.naive_longdouble) before it is written.

Inputs: float64 uniform[0,1) from ONE np.random.default_rng(0) per case, arrays
drawn in sorted-argument-name order (feinsum_amd.measure.generate_host_input_arrays;
the reference draws in hash order, measure.py:99-108, which is not reproducible).
E in {1, 7, 37}: 37 = two full 16-element MFMA tiles + a 5-element remainder.

Run from the repository root:  python tests/golden/make_golden.py
"""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import feinsum_amd as f  # noqa: E402
from feinsum_amd.measure import generate_host_input_arrays  # noqa: E402
from oracle import np_oracle  # noqa: E402


def cases():
    Np, nf, Nfp = 35, 4, 15
    yield "grad_p4", f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E")),
                              f.array("R", (3, Np, Np)), f.array("u", ("E", Np)))
    yield "div_p4", f.einsum("xre,rij,xej->ei", f.array("J", (3, 3, "E")),
                             f.array("R", (3, Np, Np)), f.array("u", (3, "E", Np)))
    yield "facemass_p4_ef_fij", f.batched_einsum(
        "ef,fij,fej->ei",
        [[f.array("J", ("E", nf)), f.array("R", (nf, Np, Nfp)), f.array(f"v{k}", (nf, "E", Nfp))]
         for k in range(4)])
    yield "facemass_p4_ifj_fe", f.batched_einsum(
        "ifj,fe,fej->ei",
        [[f.array("L", (Np, nf, Nfp)), f.array("J", (nf, "E")), f.array(f"v{k}", (nf, "E", Nfp))]
         for k in range(4)])
    yield "batched_div_p4", f.batched_einsum(
        "se,sij,ej->ei",
        [[f.array("J" + c, (3, "E")), f.array("R", (3, Np, Np)), f.array("u" + c, ("E", Np))]
         for c in "xyz"])
    yield "grad_p2", f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E")),
                              f.array("R", (3, 10, 10)), f.array("u", ("E", 10)))


def main():
    out_dir = Path(__file__).resolve().parent
    for name, expr in cases():
        for E in (1, 7, 37):
            host = generate_host_input_arrays(expr, E, np_seed=0)
            payload = {f"in_{k}": v for k, v in host.items()}
            worst = 0.0
            for out_name, row in zip(expr.output_names, expr.args):
                ops = [host[a.name] for a in row]
                ref = np_oracle.reference_outputs(expr.get_subscripts(), [ops])[0]
                chk = np_oracle.naive_longdouble(expr.get_subscripts(), ops)
                err = np_oracle.max_rel_err(ref, chk)
                assert err < 5e-15, (name, E, out_name, err)
                worst = max(worst, err)
                payload[f"out_{out_name}"] = ref
            payload["subscripts"] = np.array(expr.get_subscripts())
            np.savez(out_dir / f"{name}_E{E}.npz", **payload)
            print(f"{name} E={E}: {len(payload) - 1} arrays, oracle-vs-longdouble {worst:.2e}")


if __name__ == "__main__":
    main()
