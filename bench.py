#!/usr/bin/env python
"""
bench.py -- GFLOP/s of the DG-wave p=4 grad einsum on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload grad|div|facemass|pipeline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one evaluation of the einsum over the rank's batch of elements
(default workload: BASELINE.json configs[1], grad 'xre,rij,ej->xei', p = 4, 1e6
elements per GPU, float64, synthetic uniform[0,1) inputs resident in HBM).
Elements shard across ranks with no data-path collective (weak scaling); the
only exchange is the all-gather of the per-shard result reductions, after the
timed region.  Rank 0 prints ONE JSON line.

Measurement protocol: W untimed warm-up steps, barrier + synchronize, K timed
steps enqueued back to back on the launch stream and bracketed by HIP events on
that same stream (fe_time_launches of the C ABI), barrier + synchronize; the
step time is the MAX over ranks of the host wall-clock around the timed region.
`roofline.achieved` = algorithmic bytes per launch (grad 1192 B /
element + 29 400 B for D) / mean kernel time from the HIP events.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NP, NF, NFP, NFIELDS = 35, 4, 15, 4
SETTLE_LAUNCHES = 100          # part of setup (see main)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_GFLOPS = 78_600.0    # fp64 vector = matrix


def _einsums():
    import feinsum_amd as f

    grad = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E")), f.array("R", (3, NP, NP)),
                    f.array("u", ("E", NP)))
    div = f.einsum("xre,rij,xej->ei", f.array("J", (3, 3, "E")), f.array("R", (3, NP, NP)),
                   f.array("v", (3, "E", NP)))
    fm = f.batched_einsum(
        "ef,fij,fej->ei",
        [[f.array("Jf", ("E", NF)), f.array("L", (NF, NP, NFP)), f.array(f"w{k}", (NF, "E", NFP))]
         for k in range(NFIELDS)])
    return {"grad": [grad], "div": [div], "facemass": [fm], "graddiv": [div, grad],
            "pipeline": [div, grad, fm]}


def _device_inputs(expr, E, device, seed):
    import torch

    import feinsum_amd as f

    g = torch.Generator(device=device).manual_seed(seed)
    dev = {}
    for name in sorted(expr.all_args):
        shape = tuple(E if isinstance(d, f.SizeParam) else int(d) for d in expr.arg_to_shape[name])
        dev[name] = torch.rand(shape, dtype=torch.float64, device=device, generator=g)
    return dev


def _cpu_share() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup quota).  Oversubscribing a
    quota-limited box is several times slower (MI355X box: quota 16 CPUs, 256 visible)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_baseline(workload: str, budget_s: float = 12.0):
    """The oracle's C loop nest (optimal 2-step schedule, OpenMP) timed on the host cores."""
    import numpy as np

    from oracle import c_oracle, np_oracle

    lib_native = c_oracle.load(native=True)
    lib_native.oracle_set_num_threads(_cpu_share())   # (the OpenMP runtime is already up: torch)
    threads = int(lib_native.oracle_num_threads())
    E = 200_000
    rng = np.random.default_rng(0)
    if workload in ("grad", "graddiv", "pipeline"):
        J, D, u = rng.random((3, 3, E)), rng.random((3, NP, NP)), rng.random((E, NP))
        out = np.empty((3, E, NP))
        fn = lambda: lib_native.oracle_grad3d_hoisted(J, D, u, out, E, NP)  # noqa: E731
        flops, what = 7980.0 * E, "grad"
        check = lambda: np_oracle.max_rel_err(  # noqa: E731
            out[:, :64], np.einsum("xre,rij,ej->xei", J[:, :, :64], D, u[:64], optimize="optimal"))
    elif workload == "div":
        J, D, u = rng.random((3, 3, E)), rng.random((3, NP, NP)), rng.random((3, E, NP))
        out = np.empty((E, NP))
        fn = lambda: lib_native.oracle_div3d_hoisted(J, D, u, out, E, NP)  # noqa: E731
        flops, what = 7980.0 * E, "div"
        check = lambda: np_oracle.max_rel_err(  # noqa: E731
            out[:64], np.einsum("xre,rij,xej->ei", J[:, :, :64], D, u[:, :64], optimize="optimal"))
    else:
        J, R, v = rng.random((E, NF)), rng.random((NF, NP, NFP)), rng.random((NF, E, NFP))
        out = np.empty((E, NP))
        fn = lambda: lib_native.oracle_facemass_hoisted(J, R, v, out, E, NP, NF, NFP, 0, 0)  # noqa: E731
        flops, what = 17040.0 / NFIELDS * E, "face-mass (one field)"
        check = lambda: np_oracle.max_rel_err(  # noqa: E731
            out[:64], np.einsum("ef,fij,fej->ei", J[:64], R, v[:, :64], optimize="optimal"))
    fn()
    assert check() <= 1e-12
    t_total, reps = 0.0, 0
    while t_total < budget_s and reps < 100000:
        t0 = time.perf_counter()
        fn()
        t_total += time.perf_counter() - t0
        reps += 1
    return {
        "value": round(flops * reps / t_total * 1e-9, 2), "unit": "GFLOP/s", "cores": threads,
        "kind": "port",
        "sample": (f"oracle/loopnest.c {what} p=4 optimal 2-step schedule, gcc -O3 -march=native -fopenmp, "
                   f"{reps} x E={E} elements ({t_total:.1f} s) on {threads} threads "
                   f"(CPU share of this box; {os.cpu_count()} host CPUs visible)"),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="grad", choices=["grad", "div", "facemass", "graddiv", "pipeline"])
    ap.add_argument("--elems-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse", action="store_true",
                    help="graddiv / pipeline: one launch per einsum instead of the single fused launch (A/B)")
    args = ap.parse_args()

    import torch

    import feinsum_amd as f
    from feinsum_amd import measure, operator, parallel

    info = parallel.init_distributed()
    if info.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={info.world_size}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = torch.device("cuda", info.local_rank)
    torch.cuda.set_device(device)
    q = f.DeviceQueue(device)
    E = args.elems_per_gpu

    exprs = _einsums()[args.workload]
    stages, out_dicts, outs_all, flops_step, bytes_step, shared = [], [], [], 0.0, 0.0, {}
    for k, expr in enumerate(exprs):
        dev = _device_inputs(expr, E, device, seed=1000 * info.rank + k)
        for name in ("J", "R"):          # div and grad of one operator share J and D
            if name in dev:
                dev[name] = shared.setdefault(name, dev[name])
        outs = measure.generate_out_arrays(q, expr, E)
        stages.append((expr, dev))
        out_dicts.append(outs)
        outs_all += list(outs.values())
        flops_step += f.count_ops(expr, long_dim_length=E)
        bytes_step += measure._get_footprint_gbytes(expr, E) * 1e9
    if len(exprs) > 1:                   # J and D counted once
        bytes_step -= 8.0 * (9 * E + 3 * NP * NP)
    op = operator.bind_operator(stages, q, out_dicts=out_dicts, transform=args.variant, fuse=not args.no_fuse)

    s = q.stream_ptr
    # setup, untimed and not counted as warm-up steps: kernel attributes, first touch of every page,
    # device clocks up (a 5-step run right after allocation measured 12 % low otherwise)
    for _ in range(SETTLE_LAUNCHES):
        op.launch(s)
    torch.cuda.synchronize(device)
    for _ in range(args.warmup):
        op.launch(s)
    torch.cuda.synchronize(device)
    parallel.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    if len(op.launches) == 1 and hasattr(op.launches[0], "time_batch"):
        kernel_s = op.launches[0].time_batch(args.steps, s)    # HIP events on the launch stream
    else:
        kernel_s = op.time_batch(args.steps, s)                # same, through torch's event objects
    torch.cuda.synchronize(device)
    parallel.barrier()
    torch.cuda.synchronize(device)
    wall_s = time.perf_counter() - t0
    wall_s = parallel.max_over_ranks(wall_s, device)
    kernel_s = parallel.max_over_ranks(kernel_s, device)

    # the one exchange of the sharded path: all-gather of the per-shard result reductions
    t1 = time.perf_counter()
    gathered = parallel.allgather_reduction(parallel.result_reduction(outs_all))
    total = parallel.combine_reductions(gathered)
    torch.cuda.synchronize(device)
    allgather_ms = (time.perf_counter() - t1) * 1e3
    finite = bool(torch.isfinite(total).all().item()) and bool((total[:, 1] > 0).all().item())

    if info.rank == 0:
        n = info.world_size
        ms_per_step = wall_s / args.steps * 1e3
        value = n * flops_step / (wall_s / args.steps) * 1e-9
        kern_ms = kernel_s / args.steps * 1e3
        achieved_gbs = bytes_step / (kernel_s / args.steps) * 1e-9
        traffic = None
        tfile = ROOT / "profiles" / f"traffic_{args.workload}.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                if int(rec.get("E", -1)) == E:      # PMC bytes are per launch AT the profiled size
                    traffic = rec.get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        e_txt = f"{E:.0e}".replace("e+0", "e").replace("e+", "e")   # 1000000 -> "1e6"
        ai = flops_step / bytes_step
        roof_gflops = min(FP64_PEAK_GFLOPS, ai * HBM_PEAK_GBS)
        line = {
            "metric": f"GFLOP/s on DG-wave p=4 grad einsum ({e_txt} elems per GPU, fp64); fraction of roofline in `roofline`"
                      if args.workload == "grad" else f"GFLOP/s on DG-wave p=4 {args.workload} ({e_txt} elems per GPU, fp64)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": {"grad": f"configs[1]: grad xre,rij,ej->xei p=4 (Np=35), {e_txt} elements per GPU",
                             "div": f"div xre,rij,xej->ei p=4, {e_txt} elements per GPU",
                             "facemass": f"configs[3]: face-mass ef,fij,fej->ei x4 p=4, {e_txt} elements per GPU",
                             "graddiv": f"configs[2]: div xre,rij,xej->ei + grad sharing J and D, p=4, {e_txt} elements per GPU",
                             "pipeline": f"configs[4]: div + grad + face-mass x4, {e_txt} elements per GPU"}[args.workload],
                "elements_per_gpu": E, "parallelism": f"element-sharded x{n}, no data-path collective",
                "variant": args.variant, "device": q.device.name, "launches_per_step": list(op.entry_points),
            },
            "per_gpu_gflops": round(value / n, 1),
            "frac_of_min_roofline": round(value / n / roof_gflops, 4),
            "frac_of_fp64_peak": round(value / n / FP64_PEAK_GFLOPS, 4),
            "kernel_ms": round(kern_ms, 5),
            "result_allgather_ms": round(allgather_ms, 3), "result_finite": finite,
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_step)},
        }
        if n == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = _cpu_baseline(args.workload)
        print(json.dumps(line), flush=True)

    if info.world_size > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
