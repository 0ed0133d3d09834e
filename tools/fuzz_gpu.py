"""Seeded random sweep over families, shapes, layouts, field counts, element counts and kernel
variants against the numpy oracle (test infrastructure, like tests/).

    python tools/fuzz_gpu.py [n_cases] [seed]
"""
import random
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402
from feinsum_amd.measure import generate_host_input_arrays  # noqa: E402
from oracle import np_oracle  # noqa: E402

rng = random.Random(0)
ORDERS3 = [(4, 3), (10, 6), (20, 10), (35, 15), (56, 21), (7, 4), (13, 5)]
ORDERS2 = [(3, 2), (6, 3), (10, 4), (15, 5), (21, 6)]


def build():
    kind = rng.choice(["grad", "div", "bgrad", "bdiv", "fm", "fm_ifj", "fm_jfi", "fm_fji", "divcomp", "cross", "mass",
                       "apply", "grad2", "div2", "lift2"])
    Np, Nfp = rng.choice(ORDERS2 if kind.endswith("2") else ORDERS3)
    b = rng.choice([1, 2, 3, 4, 5, 8, 9])
    op = rng.choice(["rij", "rji"])
    if kind == "grad":
        return f"grad Np={Np} {op}", f.einsum(f"xre,{op},ej->xei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)), f.array("u", ("E", Np)))
    if kind == "div":
        return f"div Np={Np} {op}", f.einsum(f"xre,{op},xej->ei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)), f.array("u", (3, "E", Np)))
    if kind == "bgrad":
        return f"batched grad Np={Np} b={b} {op}", dg.batched_grad(b, Np, op)
    if kind == "bdiv":
        return f"batched div Np={Np} b={b} {op}", dg.batched_div(b, Np, op)
    if kind.startswith("fm"):
        fn = {"fm": dg.face_mass, "fm_ifj": dg.face_mass_ifj_fe, "fm_jfi": dg.face_mass_jfi_fe, "fm_fji": dg.face_mass_fji}[kind]
        return f"{kind} Np={Np} Nfp={Nfp} b={b}", fn(b, Np=Np, Nfp=Nfp)
    if kind == "divcomp":
        return f"div components Np={Np}", dg.batched_div_components(Np)
    if kind == "cross":
        return f"cross product Np={Np} {op}", dg.cross_product_batch(Np, op)
    if kind == "mass":
        return f"mass apply Np={Np} b={b}", dg.mass_apply(b, Np, rng.choice(["ij", "ji"]))
    if kind == "apply":
        return f"operator apply Np={Np}", dg.operator_apply(Np, rng.choice(["ij", "ji"]))
    if kind == "grad2":
        return f"2D grad Np={Np} b={b}", f.batched_einsum(f"xre,{op},ej->xei", [[f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array(f"u{k}", ("E", Np))] for k in range(b)])
    if kind == "div2":
        return f"2D div Np={Np} b={b}", f.batched_einsum(f"xre,{op},xej->ei", [[f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np)), f.array(f"u{k}", (2, "E", Np))] for k in range(b)])
    return f"2D lift Np={Np} Nfp={Nfp} b={b}", f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)), f.array(f"v{k}", (3, "E", Nfp))] for k in range(b)])


def run(n_cases: int, seed: int) -> int:
    """Number of failing (case, variant, output) triples; prints them."""
    rng.seed(seed)
    worst, failures = 0.0, 0
    for case in range(n_cases):
        name, expr = build()
        E = rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 500, 1003, 4099, rng.randrange(1, 30000)])
        host = generate_host_input_arrays(expr, E, np_seed=case)
        ref = {n: np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in row]])[0]
               for n, row in zip(expr.output_names, expr.args)}
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        for variant in ("auto", "generic", "mfma", "tiled"):
            try:
                outs = f.evaluate(expr, 0, dev, transform=variant, wait=True)
            except NotImplementedError:
                continue
            for k in ref:
                got = outs[k].cpu().numpy()
                err = np_oracle.max_rel_err(got, ref[k]) if ref[k].size else 0.0
                worst = max(worst, err)
                if got.shape != ref[k].shape or not np.isfinite(got).all() or err > 1e-12:
                    failures += 1
                    print(f"FAIL case {case}: {name} E={E} variant={variant} output {k}: err {err:.3e}", flush=True)
    print(f"{n_cases} cases, worst relative error {worst:.3e}, failures {failures}")
    return failures





def run_operator(n_cases: int, seed: int) -> int:
    """div + grad (+ lift) bound as one operator: fused launches against the oracle."""
    rng.seed(seed)
    failures = 0
    for case in range(n_cases):
        Np, Nfp = rng.choice(ORDERS3[:5])
        b = rng.choice([1, 2, 3, 4, 5])
        E = rng.choice([1, 15, 16, 17, 47, 48, 49, 79, 80, 81, 1003, rng.randrange(1, 20000)])
        exprs = [dg.div(Np), dg.grad(Np)] + ([rng.choice([dg.face_mass, dg.face_mass_ifj_fe])(b, Np=Np, Nfp=Nfp)]
                                            if rng.random() < 0.7 else [])
        hosts = [generate_host_input_arrays(e, E, np_seed=case + k) for k, e in enumerate(exprs)]
        if rng.random() < 0.8:
            hosts[1]["J"], hosts[1]["R"] = hosts[0]["J"], hosts[0]["R"]
        devs = [{k: torch.from_numpy(v).cuda() for k, v in h.items()} for h in hosts]
        if hosts[1]["J"] is hosts[0]["J"]:
            devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
        outs = f.evaluate_operator(list(zip(exprs, devs)), 0, wait=True)
        for e, h, o in zip(exprs, hosts, outs):
            for name, row in zip(e.output_names, e.args):
                ref = np_oracle.reference_outputs(e.get_subscripts(), [[h[a.name] for a in row]])[0]
                got = o[name].cpu().numpy()
                if not np.isfinite(got).all() or np_oracle.max_rel_err(got, ref) > 1e-12:
                    failures += 1
                    print(f"FAIL operator case {case}: Np={Np} b={b} E={E} {e.get_subscripts()} {name}", flush=True)
    print(f"{n_cases} operator cases, failures {failures}")
    return failures


def run_einsum(n_cases: int, seed: int) -> int:
    """Random explicit-mode einsums through the generic kernels (lane groups, pointwise stream)."""
    rng.seed(seed)
    failures = 0
    for case in range(n_cases):
        letters = "abcde"
        dims = {c: rng.choice([1, 2, 3, 4, 7, 16, 35]) for c in letters}
        long_idx = rng.choice(letters)
        n_ops = rng.choice([1, 2, 3])
        ops = ["".join(rng.sample(letters, rng.choice([1, 2, 3]))) for _ in range(n_ops)]
        if not any(long_idx in o for o in ops):
            ops[0] = long_idx + "".join(c for c in ops[0] if c != long_idx)[:2]
        used = sorted(set("".join(ops)))
        out = "".join(c for c in rng.sample(used, len(used)) if c == long_idx or rng.random() < 0.5)
        if long_idx not in out:
            out += long_idx
        subs = ",".join(ops) + "->" + out
        arrays = [f.array(f"A{k}", tuple("E" if c == long_idx else dims[c] for c in o)) for k, o in enumerate(ops)]
        try:
            expr = f.einsum(subs, *arrays)
        except (ValueError, TypeError):
            continue
        E = rng.choice([1, 5, 64, 257, 3000])
        host = generate_host_input_arrays(expr, E, np_seed=case)
        ref = np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in expr.args[0]]])[0]
        got = f.evaluate(expr, 0, {k: torch.from_numpy(v).cuda() for k, v in host.items()}, wait=True)["_fe_out"].cpu().numpy()
        if got.shape != ref.shape or not np.isfinite(got).all() or np_oracle.max_rel_err(got, ref) > 1e-12:
            failures += 1
            print(f"FAIL einsum case {case}: {subs} dims={dims} E={E}", flush=True)
    print(f"{n_cases} einsum cases, failures {failures}")
    return failures


def run_dynamic_walk(n_cases: int, seed: int) -> int:
    """Launches large enough for the dynamic walk (five or more rounds of tiles: fe_common.h) -- single launches of all
    tetrahedral orders, batched launches, triangles, fused operators at random element counts: tickets against the static
    walk, bitwise, and the ticket launch a second time (its counters must have been left zeroed).  Round 4: on a random stream
    (the default one or a fresh one: the counters belong to the stream), with a random grid size (fe_set_cu_limit: grids of
    fewer than 128 blocks must walk statically) and a random load mode (fe_set_temporal_loads_mib)."""
    from feinsum_amd import _hip

    rng.seed(seed)
    failures = 0
    before = _hip.set_tail_rounds(1 << 20)
    before_cus, before_mib = _hip.set_cu_limit(0), _hip.set_temporal_loads_mib(248)
    try:
        for case in range(n_cases):
            Np, Nfp = rng.choice(ORDERS3[:5])
            E = rng.randrange(170_000 if Np >= 20 else 900_000, 700_000 if Np >= 20 else 1_300_000)
            kind = rng.choice(["grad", "div", "fm3", "fm4", "fm5", "bgrad", "bdiv", "grad2", "div2", "lift2", "operator"])
            if kind == "operator":
                exprs = [dg.div(Np), dg.grad(Np), dg.face_mass(rng.choice([2, 3, 4]), Np=Np, Nfp=Nfp)]
            elif kind.endswith("2"):
                Np, Nfp = rng.choice(ORDERS2)
                E = rng.randrange(1_400_000, 1_800_000)
                J, R = f.array("J", (2, 2, "E")), f.array("R", (2, Np, Np))
                b = rng.choice([1, 2, 3])
                exprs = [{"grad2": lambda: f.batched_einsum("xre,rij,ej->xei", [[J, R, f.array(f"u{k}", ("E", Np))] for k in range(b)]),
                          "div2": lambda: f.batched_einsum("xre,rij,xej->ei", [[J, R, f.array(f"u{k}", (2, "E", Np))] for k in range(b)]),
                          "lift2": lambda: f.batched_einsum("ef,fij,fej->ei", [[f.array("J", ("E", 3)), f.array("R", (3, Np, Nfp)),
                                                                               f.array(f"v{k}", (3, "E", Nfp))] for k in range(3)])}[kind]()]
            else:
                b = rng.choice([2, 3])
                exprs = [{"grad": lambda: dg.grad(Np), "div": lambda: dg.div(Np), "fm3": lambda: dg.face_mass(3, Np=Np, Nfp=Nfp),
                          "fm4": lambda: dg.face_mass(4, Np=Np, Nfp=Nfp), "fm5": lambda: dg.face_mass(5, Np=Np, Nfp=Nfp),
                          "bgrad": lambda: dg.batched_grad(b, Np), "bdiv": lambda: dg.batched_div(b, Np)}[kind]()]
            gen = torch.Generator(device="cuda").manual_seed(case)
            devs = [{a: torch.rand(tuple(E if isinstance(d, f.SizeParam) else int(d) for d in e.arg_to_shape[a]), dtype=torch.float64,
                                   device="cuda", generator=gen) for a in sorted(e.all_args)} for e in exprs]
            if len(exprs) == 3:
                devs[1]["J"], devs[1]["R"] = devs[0]["J"], devs[0]["R"]
            q = f.DeviceQueue(0, stream=torch.cuda.Stream()) if rng.random() < 0.5 else f.DeviceQueue(0)
            torch.cuda.synchronize()

            def evaluate():
                if len(exprs) == 1:
                    return [f.evaluate(exprs[0], q, devs[0], wait=True)]
                return f.evaluate_operator(list(zip(exprs, devs)), q, wait=True)
            _hip.set_tail_rounds(-1)
            _hip.set_cu_limit(0)
            _hip.set_temporal_loads_mib(0)
            static = [{k: v.clone() for k, v in o.items()} for o in evaluate()]
            rounds, cus, mib = rng.choice([1 << 20, 1 << 20, 1, 4]), rng.choice([0, 0, 0, 32, 64, 100, 160]), rng.choice([0, 248, 1 << 20])
            _hip.set_tail_rounds(rounds)
            _hip.set_cu_limit(cus)
            _hip.set_temporal_loads_mib(mib)
            for rep in range(2):
                for o_static, o in zip(static, evaluate()):
                    for k in o_static:
                        if not torch.equal(o_static[k], o[k]):
                            failures += 1
                            print(f"FAIL dynamic walk case {case}: {kind} Np={Np} E={E} rounds={rounds} cus={cus} loads<={mib}MiB output {k} (launch {rep})",
                                  flush=True)
            del static, devs
        dirty = _hip.tail_check()["dirty_words"]
        if dirty:
            failures += 1
            print(f"FAIL: {dirty} ticket-counter words left non-zero", flush=True)
    finally:
        _hip.set_tail_rounds(before)
        _hip.set_cu_limit(before_cus)
        _hip.set_temporal_loads_mib(before_mib)
    print(f"{n_cases} dynamic-walk cases, failures {failures}")
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n, seed) + run_operator(n // 4, seed) + run_einsum(n, seed) + run_dynamic_walk(n // 5, seed) else 0)
