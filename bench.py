#!/usr/bin/env python
"""
bench.py -- GFLOP/s of the DG-wave p=4 grad einsum on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload grad|div|facemass|graddiv|pipeline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one evaluation of the einsum over the rank's batch of elements
(default workload: BASELINE.json configs[1], grad 'xre,rij,ej->xei', p = 4, 1e6
elements per GPU, float64, synthetic uniform[0,1) inputs resident in HBM).
Elements shard across ranks with no data-path collective (weak scaling by
default; ``--elems-total`` splits one global batch instead); the only exchange
is the all-gather of the per-shard result reductions, after the timed region.
Rank 0 prints ONE JSON line.

Measurement protocol: ``setup_launches`` untimed launches over ``setup_seconds``
(first touch of every page, kernel attributes, device clocks and power state
settled -- both reported in the line), W
untimed warm-up steps, barrier + synchronize, K timed steps enqueued back to
back on the launch stream and bracketed by HIP events on that same stream
(fe_time_launches of the C ABI), barrier + synchronize; the step time is the
MAX over ranks of the host wall-clock around the timed region.
``roofline.achieved`` = algorithmic bytes per launch (grad 1192 B /
element + 29 400 B for D) / mean kernel time from the HIP events.
``protocol_ms_per_step`` is the same launch timed with the reference's own
protocol (5 warm-ups, batches of 5, >= 10 launches and >= 2 s;
src/feinsum/measure.py:248-275) right after the timed region.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NP, NF, NFP, NFIELDS = 35, 4, 15, 4
SETUP_LAUNCHES = 100           # untimed, before the W warm-up steps; reported as `setup_launches` ...
SETUP_SECONDS = 0.5            # ... and continued until this much time has passed (`setup_seconds`): see main()
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_GFLOPS = 78_600.0    # fp64 vector = matrix
WORKLOADS = ("grad", "div", "facemass", "graddiv", "pipeline")


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


# --------------------------------------------------------------------------
# rank logic (CPU-testable: tests/test_bench_ranks_cpu.py drives it with gloo)
# --------------------------------------------------------------------------

def rank_elements(args, info) -> int:
    """Elements this rank evaluates: ``--elems-per-gpu`` each (weak scaling, the default), or this
    rank's tile-aligned block of ``--elems-total`` (the last rank takes the ragged tail)."""
    from feinsum_amd import parallel

    if args.elems_total:
        start, stop = parallel.shard_bounds(args.elems_total, info.world_size, info.rank)
        return stop - start
    return args.elems_per_gpu


def timed_region(step_batch, steps: int, sync, device=None, keep_busy=None):
    """barrier + sync | K steps | sync | barrier + sync.  *step_batch(n)* enqueues n steps and returns their device
    seconds.  *keep_busy()* (a few untimed launches + synchronize) is what a rank does while it waits in the OPENING barrier for
    the others, so that no rank times its steps on a device that has just been idling (parallel.barrier_keeping_busy).
    Returns a dict of seconds:

    ``wall``            MAX over ranks of each rank's own [after the opening barrier, its own K steps done and synchronised]
                        -- the sharded path has no data-path collective, so this is when the job's work is done;
    ``wall_barrier``    MAX over ranks of the same interval extended over the closing barrier + synchronize (one more
                        collective: 50-200 us of RCCL all-reduce and Python, 1-5 % of a 3.8 ms region -- reported, not charged
                        to the kernels);
    ``kernel``          MAX over ranks of the device seconds of the K steps (HIP events on the launch stream);
    ``local_wall`` / ``local_kernel``   this rank's own two figures.
    """
    from feinsum_amd import parallel

    sync()
    busy_calls = parallel.barrier_keeping_busy(keep_busy)
    sync()
    t0 = time.perf_counter()
    kernel_s = step_batch(steps)
    sync()
    local_s = time.perf_counter() - t0
    parallel.barrier()
    sync()
    closed_s = time.perf_counter() - t0
    return {"wall": parallel.max_over_ranks(local_s, device), "wall_barrier": parallel.max_over_ranks(closed_s, device),
            "kernel": parallel.max_over_ranks(kernel_s, device), "local_wall": local_s, "local_kernel": kernel_s,
            "opening_barrier_busy_calls": busy_calls}


_PLACEMENT_MODES = ("split", "separate", "separate (split allocator failed)")


def gather_rank_reports(report: dict, device=None) -> list:
    """Every rank's own figures in rank 0's line (one all-gather of a few doubles): elements, kernel and wall milliseconds
    per step of ITS timed steps, and what its allocator did -- `value` is the job's flops over the SLOWEST rank's time, so a
    rank on unsplit arrays or with a ten-second search must be attributable.  *report*: ``elements``, ``kernel_ms``,
    ``wall_ms``, ``placement_mode`` (one of _PLACEMENT_MODES), ``unsplit_arrays``, ``allocator_ms``."""
    from feinsum_amd import parallel

    mode = report.get("placement_mode", "separate")
    row = [report["elements"], report["kernel_ms"], report["wall_ms"], _PLACEMENT_MODES.index(mode) if mode in _PLACEMENT_MODES else 1,
           report.get("unsplit_arrays", 0), report.get("allocator_ms", 0.0), report.get("search_ms", 0.0), report.get("release_ms", 0.0),
           report.get("busy_calls", 0)]
    out = []
    for rank, r in enumerate(parallel.gather_rows(row, device)):
        out.append({"rank": rank, "elements": int(r[0]), "kernel_ms": round(r[1], 5), "wall_ms": round(r[2], 5),
                    "placement_mode": _PLACEMENT_MODES[int(r[3])], "unsplit_arrays": int(r[4]), "allocator_ms": round(r[5], 1),
                    "allocator_search_ms": round(r[6], 1), "allocator_release_ms": round(r[7], 1),
                    # (how long this rank kept its device busy in the opening barrier while the others arrived: calls of ~1 ms)
                    "opening_barrier_busy_calls": int(r[8]) if len(r) > 8 else 0})
    return out


def exchange_results(outs, sync):
    """The one exchange of the sharded path: every rank reduces its outputs locally
    ([sum, sum of squares, max |.|] per output), the few doubles are all-gathered and combined.
    Returns (combined [n_outputs, 3] tensor, local-reduction ms, all-gather ms): the first time is
    torch reductions over this rank's own outputs, only the second is a collective."""
    from feinsum_amd import parallel

    sync()
    t0 = time.perf_counter()
    local = parallel.result_reduction(outs)
    sync()
    t1 = time.perf_counter()
    gathered = parallel.allgather_reduction(local)
    total = parallel.combine_reductions(gathered)
    sync()
    t2 = time.perf_counter()
    return total, (t1 - t0) * 1e3, (t2 - t1) * 1e3


def compose_line(*, workload, n_gpus, steps, warmup, setup_launches=SETUP_LAUNCHES, wall_s, kernel_s, flops_step_all, flops_step_rank0,
                 bytes_step_rank0, elems_rank0, elems_total, variant, device_name, entry_points, extra):
    """The JSON line from measured quantities (pure)."""
    ms_per_step = wall_s / steps * 1e3
    value = flops_step_all / (wall_s / steps) * 1e-9            # whole job: all ranks' flops / max-over-ranks time
    achieved_gbs = bytes_step_rank0 / (kernel_s / steps) * 1e-9
    e_txt = f"{elems_rank0:.0e}".replace("e+0", "e").replace("e+", "e")   # 1000000 -> "1e6"
    ai = flops_step_rank0 / bytes_step_rank0
    roof_gflops = min(FP64_PEAK_GFLOPS, ai * HBM_PEAK_GBS)
    names = {"grad": f"configs[1]: grad xre,rij,ej->xei p=4 (Np=35), {e_txt} elements per GPU",
             "div": f"div xre,rij,xej->ei p=4, {e_txt} elements per GPU",
             "facemass": f"configs[3]: face-mass ef,fij,fej->ei x4 p=4, {e_txt} elements per GPU",
             "graddiv": f"configs[2]: div xre,rij,xej->ei + grad sharing J and D in one launch (fusion buys the launch "
                        f"boundary only; J is fetched by both bodies), p=4, {e_txt} elements per GPU",
             "pipeline": f"configs[4]: div + grad + face-mass x4, {e_txt} elements per GPU"}
    line = {
        "metric": (f"GFLOP/s on DG-wave p=4 grad einsum ({e_txt} elems per GPU, fp64); fraction of roofline in `roofline`"
                   if workload == "grad" else f"GFLOP/s on DG-wave p=4 {workload} ({e_txt} elems per GPU, fp64)"),
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": n_gpus, "steps": steps,
        "warmup": warmup, "setup_launches": setup_launches, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong" if elems_total else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": names[workload],
            "elements_per_gpu": elems_rank0, "elements_total": elems_total or elems_rank0 * n_gpus,
            "parallelism": f"element-sharded x{n_gpus}, no data-path collective",
            "variant": variant, "device": device_name, "launches_per_step": list(entry_points),
        },
        "per_gpu_gflops": round(value / n_gpus, 1),
        "frac_of_min_roofline": round(value / n_gpus / roof_gflops, 4),
        "frac_of_fp64_peak": round(value / n_gpus / FP64_PEAK_GFLOPS, 4),
        "kernel_ms": round(kernel_s / steps * 1e3, 5),
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": int(bytes_step_rank0)},
    }
    line.update(extra)
    return line


# --------------------------------------------------------------------------
# evidence that is not measured live: PMC counters of a committed profile
# --------------------------------------------------------------------------

def strip_c_comments(text: str) -> str:
    """*text* (C / C++ / HIP source) without comments and with every run of white space collapsed: what the
    compiler sees.  String and character literals are kept as they are."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            j = text.find("\n", i)
            while j > 0 and text[j - 1] == "\\":          # a line comment continued by a backslash
                j = text.find("\n", j + 1)
            i = n if j < 0 else j
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_sha() -> str:
    """Hash of the kernel sources the library is built from, AS THE COMPILER SEES THEM (comments and white space
    removed: an edit to a comment leaves the binary -- and so a committed profile of it -- valid).  There is no .git
    on the GPU box, so this -- not a commit hash -- ties a committed profile to the binary that is running."""
    h = hashlib.sha256()
    for p in sorted((ROOT / "feinsum_amd" / "csrc").glob("*")) + [ROOT / "include" / "feinsum_hip.h"]:
        if p.suffix not in (".h", ".hip", ".cpp", ".hpp"):
            continue
        h.update(p.name.encode())
        h.update(strip_c_comments(p.read_text(errors="replace")).encode())
    # ... and how they are compiled: the flags of __graft_entry__.build_library and the compiler's version (round 5: another
    # optimisation level or another hipcc is another binary, whatever the sources say)
    h.update(("flags:" + " ".join(_build_flags())).encode())
    h.update(("compiler:" + _compiler_version()).encode())
    return h.hexdigest()[:16]


def _build_info() -> dict:
    """feinsum_amd/libfeinsum_hip.build.json, written by __graft_entry__.build_library when it compiles the library: the flags
    and the compiler it was built with.  Read from the file -- never asked of the compiler here: bench.py has initialised the
    GPU by the time it needs the hash (under rocprofv3 --pmc the profiler has, before the program starts), and such a process
    must not start another program."""
    try:
        return json.loads((ROOT / "feinsum_amd" / "libfeinsum_hip.build.json").read_text())
    except (OSError, ValueError):
        return {}


def _build_flags():
    return list(_build_info().get("hipcc_flags") or ["unknown"])


def _compiler_version() -> str:
    return str(_build_info().get("compiler") or "unknown")


def committed_counters(workload: str, E: int):
    """HBM bytes per launch and MFMA utilisation from profiles/traffic_<workload>.json (rocprofv3
    --pmc passes, tools/pmc_summary.py) -- only if that profile was taken at this element count on
    a library built from exactly these kernel sources; otherwise (None, note)."""
    tfile = ROOT / "profiles" / (f"traffic_{workload}.json" if E == 1_000_000 else f"traffic_{workload}_E{E}.json")
    try:
        rec = json.loads(tfile.read_text())
    except (OSError, ValueError):
        return None, "no committed PMC profile for this workload"
    if int(rec.get("E", -1)) != E:
        return None, f"committed PMC profile is for E={rec.get('E')}"
    if rec.get("source_sha") != kernel_source_sha():
        return None, (f"committed PMC profile was taken on kernel sources {rec.get('source_sha')}, "
                      f"this library is {kernel_source_sha()}")
    return rec, None


# --------------------------------------------------------------------------
# device state (clocks / power / temperatures) around the timed region
# --------------------------------------------------------------------------

def device_sampler(ordinal: int = 0):
    """tools/device_state.Sampler for HIP device *ordinal* (matched to its sysfs card by PCI address), or a
    do-nothing stand-in: diagnostics never fail the bench."""
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import torch

        import device_state as ds

        pr = torch.cuda.get_device_properties(ordinal)
        pci = None
        if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            pci = (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        return ds.Sampler(ordinal, pci)
    except Exception as exc:   # noqa: BLE001
        class _Null:
            def __enter__(self):
                return self

            def __exit__(self, *a):
                return None

            def summary(self, _e=str(exc)[:200]):
                return {"error": _e}

        return _Null()


# --------------------------------------------------------------------------
# CPU baseline (test infrastructure timed beside the GPU number)
# --------------------------------------------------------------------------

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


def _cpu_model() -> str:
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _reference_protocol(fn, min_secs=2.0, warmups=5, batch=5, min_rounds=10, max_secs=None):
    """Seconds per call with the timing loop of src/feinsum/measure.py:248-275 (host clock).
    *max_secs*: stop after the batch that crosses it (bounded sample; the line says so)."""
    for _ in range(warmups):
        fn()
    total, rounds = 0.0, 0
    while rounds < min_rounds or total < min_secs:
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        total += time.perf_counter() - t0
        rounds += batch
        if max_secs is not None and total >= max_secs:
            break
    return total / rounds, rounds


def cpu_baseline(workload: str, budget_s: float = 8.0, full: bool = False):
    """The oracle's C loop nests and the reference's own ground-truth evaluator on the host cores.

    Headline `value`: the hoisted (optimal 2-step schedule) C nest, OpenMP over the CPU share.
    `also`: the two stand-ins SURVEY §8(d) asks for, at E = 1e4 (BASELINE configs[0]) and E = 1e6:
    (i) ``np.einsum(optimize="optimal")`` -- what the reference validates against
    (src/feinsum/measure.py:149-159) -- and (ii) the trivial single-statement nest, what a CPU
    OpenCL device runs under the identity transform of test/test_codegen.py:115-120; both with the
    reference's timing protocol, cut short at E = 1e6 unless *full* (stated per entry)."""
    import numpy as np

    from oracle import c_oracle, np_oracle

    lib = c_oracle.load(native=True)
    lib.oracle_set_num_threads(_cpu_share())   # (the OpenMP runtime is already up: torch)
    threads = int(lib.oracle_num_threads())
    rng = np.random.default_rng(0)
    subs = {"grad": "xre,rij,ej->xei", "div": "xre,rij,xej->ei", "facemass": "ef,fij,fej->ei"}
    fam = workload if workload in subs else "grad"          # graddiv / pipeline: the grad einsum

    def operands(E):
        if fam == "grad":
            return [rng.random((3, 3, E)), rng.random((3, NP, NP)), rng.random((E, NP))], np.empty((3, E, NP))
        if fam == "div":
            return [rng.random((3, 3, E)), rng.random((3, NP, NP)), rng.random((3, E, NP))], np.empty((E, NP))
        return [rng.random((E, NF)), rng.random((NF, NP, NFP)), rng.random((NF, E, NFP))], np.empty((E, NP))

    def c_call(kind, ops, out, E):
        if fam == "grad":
            return lambda: getattr(lib, f"oracle_grad3d_{kind}")(*ops, out, E, NP)
        if fam == "div":
            return lambda: getattr(lib, f"oracle_div3d_{kind}")(*ops, out, E, NP)
        return lambda: getattr(lib, f"oracle_facemass_{kind}")(*ops, out, E, NP, NF, NFP, 0, 0)

    flops_elem = 7980.0 if fam != "facemass" else 17040.0 / NFIELDS
    what = {"grad": "grad", "div": "div", "facemass": "face-mass (one field)"}[fam]

    # ---- headline: hoisted nest, bounded sample
    E = 200_000
    ops, out = operands(E)
    fn = c_call("hoisted", ops, out, E)
    fn()
    n = 64                                        # the checker: first 64 elements against np.einsum
    if fam == "grad":
        head, got = [ops[0][:, :, :n], ops[1], ops[2][:n]], out[:, :n]
    elif fam == "div":
        head, got = [ops[0][:, :, :n], ops[1], ops[2][:, :n]], out[:n]
    else:
        head, got = [ops[0][:n], ops[1], ops[2][:, :n]], out[:n]
    ref = np.einsum(subs[fam], *head, optimize="optimal")
    assert np_oracle.max_rel_err(got, ref) <= 1e-12
    t_total, reps = 0.0, 0
    while t_total < budget_s and reps < 100000:
        t0 = time.perf_counter()
        fn()
        t_total += time.perf_counter() - t0
        reps += 1
    result = {
        "value": round(flops_elem * E * reps / t_total * 1e-9, 2), "unit": "GFLOP/s", "cores": threads,
        "kind": "port", "cpu_model": _cpu_model(),
        "sample": (f"oracle/loopnest.c {what} p=4 optimal 2-step schedule, gcc -O3 -march=native -fopenmp, "
                   f"{reps} x E={E} elements ({t_total:.1f} s) on {threads} threads "
                   f"(CPU share of this box; {os.cpu_count()} host CPUs visible)"),
        "also": [],
    }

    # ---- SURVEY §8(d) stand-ins at E = 1e4 and E = 1e6
    for E in (10_000, 1_000_000):
        ops, out = operands(E)
        big = E >= 1_000_000 and not full
        ein = lambda: np.einsum(subs[fam], *ops, optimize="optimal")   # noqa: E731
        if big:      # ~5-8 s per call: the full protocol (15+ calls) is `--cpu-baseline full`
            s, n = _reference_protocol(ein, warmups=1, batch=1, min_rounds=2, min_secs=0.0)
            proto = "shortened: 1 warm-up + 2 calls (full protocol: bench.py --cpu-baseline full, profiles/r02/cpu_baseline_full.json)"
        else:
            s, n = _reference_protocol(ein)
            proto = "reference protocol: 5 warm-ups, batches of 5, >= 10 calls and >= 2 s"
        result["also"].append({"evaluator": 'np.einsum(optimize="optimal")', "E": E, "threads": "numpy/BLAS default",
                               "gflops": round(flops_elem * E / s * 1e-9, 3), "seconds_per_call": round(s, 6),
                               "calls": n, "protocol": proto})
        triv = c_call("trivial", ops, out, E)
        s, n = _reference_protocol(triv, max_secs=6.0 if big else None)
        entry = {"evaluator": "oracle/loopnest.c trivial single-statement nest (identity transform), OpenMP", "E": E,
                 "threads": threads, "gflops": round(flops_elem * E / s * 1e-9, 3),
                 "seconds_per_call": round(s, 6), "calls": n,
                 "protocol": "reference protocol" + (", stopped after 6 s" if big else "")}
        if fam == "grad":
            entry["executed_gflops"] = round(33075.0 * E / s * 1e-9, 2)   # test/test_loopy_utils.py:270: 33075 per element
        result["also"].append(entry)
    return result


# --------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks
# --------------------------------------------------------------------------

def needs_own_ranks(gpus: int, environ) -> bool:
    """True when this process was started plainly (no RANK / WORLD_SIZE from a launcher) but asked for several GPUs:
    it then becomes the parent of N rank processes and touches no GPU itself."""
    return gpus > 1 and "RANK" not in environ and "WORLD_SIZE" not in environ


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(n: int, argv, *, script=None, environ=None, timeout_s: float = 3000.0, out=None, err=None) -> int:
    """
    Start *n* fresh rank processes of this script (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, as torch.distributed.run sets them), relay rank 0's standard output (the JSON
    line), send the other ranks' standard output to our standard error, and return 0 only if every rank exited with 0.
    The first rank that fails (or the timeout) ends the others -- by the PIDs started here, never by pattern.  The
    parent makes no GPU call and replaces no process image: the ranks are children.
    """
    import subprocess
    import threading

    out = out or sys.stdout
    err = err or sys.stderr
    env0 = dict(os.environ if environ is None else environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env0["FEINSUM_BENCH_LAUNCHER"] = "bench.py (self-spawned ranks)"
    script = str(script or Path(__file__).resolve())
    procs = []
    for rank in range(n):
        env = dict(env0, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stdout=subprocess.PIPE, stderr=None, text=True))

    def relay(proc, sink):
        for line in proc.stdout:
            sink.write(line)
            sink.flush()

    threads = [threading.Thread(target=relay, args=(p_, out if r == 0 else err), daemon=True) for r, p_ in enumerate(procs)]
    for t in threads:
        t.start()
    deadline = time.monotonic() + timeout_s
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p_ in enumerate(procs):
            if codes[r] is None:
                codes[r] = p_.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = (r, codes[r])
        if failed is not None or time.monotonic() > deadline:
            break
        time.sleep(0.05)
    if any(c is None for c in codes):      # a rank failed or the time ran out: end the ranks that are still running
        why = f"rank {failed[0]} exited with {failed[1]}" if failed else f"no result after {timeout_s:.0f} s"
        err.write(f"bench.py: {why}; stopping the other ranks\n")
        for r, p_ in enumerate(procs):
            if codes[r] is None:
                p_.terminate()
        for r, p_ in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = p_.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p_.kill()
                    codes[r] = p_.wait()
    for t in threads:
        t.join(timeout=5)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        err.write(f"bench.py: ranks with a non-zero exit code: {bad}\n")
        return 1
    return 0


# --------------------------------------------------------------------------

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="grad", choices=WORKLOADS)
    ap.add_argument("--elems-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--elems-total", type=int, default=0,
                    help="split ONE global batch of this many elements over the ranks (strong scaling) instead of "
                         "--elems-per-gpu each")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="bounded", choices=["bounded", "full"],
                    help="full: the complete reference timing protocol for the E = 1e6 stand-ins too (~2-3 min)")
    ap.add_argument("--no-protocol", action="store_true", help="skip the reference-protocol timing (2 s)")
    ap.add_argument("--setup-seconds", type=float, default=SETUP_SECONDS,
                    help="untimed launches continue until this many seconds have passed (reported in the line): the device's "
                         "memory clocks and power state settle over the first ~0.3 s of load")
    ap.add_argument("--setup-launches", type=int, default=SETUP_LAUNCHES,
                    help="untimed launches before the warm-up steps (reported in the line)")
    ap.add_argument("--prepare", action="store_true",
                    help="prepare the operator matrices once at bind time (fe_prepare_operator) instead of rebuilding the "
                         "MFMA fragments in every launch; measured: no gain, see DESIGN.md")
    ap.add_argument("--placement", default="split", choices=["split", "separate"],
                    help="split (default): one allocation per array, the OUTPUTS from the split allocator "
                         "(feinsum_amd.placement.zeros: 4 MiB pieces alternating between two classes of physical memory; no arena, no scan); "
                         "separate: every array from torch")
    ap.add_argument("--no-fuse", action="store_true",
                    help="graddiv / pipeline: one launch per einsum instead of the single fused launch (A/B)")
    ap.add_argument("--gather-fields", choices=("on", "off"), default="off",
                    help="on: after rank 0 has printed the JSON line, all-gather every output field over the ranks and report "
                         "the rate on stderr (SURVEY 8e's optional full-field exchange; never part of `value`)")
    ap.add_argument("--spawn-timeout", type=float, default=3000.0,
                    help="--gpus N started without a launcher: seconds the N rank processes may take")
    args = ap.parse_args()

    # `python bench.py --gpus N` without RANK / WORLD_SIZE: this process becomes the parent of N ranks (no torch.cuda, no
    # HIP call here) and exits with their verdict; under torch.distributed.run the ranks arrive with their environment set
    if needs_own_ranks(args.gpus, os.environ):
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:], timeout_s=args.spawn_timeout))
    # several ranks start their allocator searches at the same instant, each on its own device but through one driver: bound
    # what a search for a second class of memory may take (default 4 s; the arrays are then of one class and say so --
    # `placement.pool.unsplit_arrays` in the line -- and run like ordinary allocations)
    # (round 5: the budget is the pool's TOTAL, wall clock, release of the skipped memory included -- fe_split_alloc.h; round 4's
    # bounded the spacer creation of one search and let a 10 s search through in the four-rank rehearsal)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("FEINSUM_SPLIT_SEARCH_MS", "2500")
        os.environ.setdefault("FEINSUM_SPLIT_SEARCH_GIB", "32")
        try:       # ranks that share a device (rehearsals) search their share of its free memory only; counting devices does not
            import torch as _t   # initialise the GPU

            n_dev = max(_t.cuda.device_count(), 1)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
            os.environ.setdefault("FEINSUM_SPLIT_SHARE", str(max(1, -(-local_world // n_dev))))
        except Exception:   # noqa: BLE001
            pass

    # a kernel that compiled to fewer resident blocks per CU than its launch geometry assumes is an error here, not a
    # warning (feinsum_hip.hip: configure_kernel): a number taken at half the residency is not the product's
    os.environ.setdefault("FEINSUM_STRICT_RESIDENCY", "1")

    import torch

    import feinsum_amd as f
    from feinsum_amd import _hip, measure, operator, parallel

    info = parallel.init_distributed()
    if info.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {info.world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = torch.device("cuda", info.local_rank)
    torch.cuda.set_device(device)
    # the first collective creates the communicator (RCCL: hundreds of milliseconds): here, not in front of the timed region
    parallel.barrier()
    q = f.DeviceQueue(device)
    E = rank_elements(args, info)
    sync = lambda: torch.cuda.synchronize(device)   # noqa: E731

    exprs = _einsums()[args.workload]
    flops_step = sum(float(f.count_ops(expr, long_dim_length=E)) for expr in exprs)
    bytes_step = sum(measure._get_footprint_gbytes(expr, E) * 1e9 for expr in exprs)
    if len(exprs) > 1:                   # J and D counted once
        bytes_step -= 8.0 * (9 * E + 3 * NP * NP)

    # (div's two-window walk, FE_VARIANT_MFMA_SPLIT, is an explicit choice: --variant mfma_split; it pays when the output
    # array is cut across two classes of physical memory in the middle, which is how the split allocator lays arrays out)
    variant = args.variant

    def bind(stages, out_dicts, variant_=None):
        return operator.bind_operator(stages, q, out_dicts=out_dicts, transform=variant_ or variant, fuse=not args.no_fuse,
                                      prepare=args.prepare)

    def separate_allocations(split=False):
        """One allocation per array: inputs from torch, outputs from torch (what round 1 measured) or, *split*, from
        the split allocator."""
        stages, out_dicts, shared = [], [], {}
        if split:   # the outputs of all stages are allocated one after the other: tell the allocator what is coming
            from feinsum_amd import placement as _placement

            total = sum(8 * len(expr.output_names) * math.prod(E if isinstance(d, f.SizeParam) else int(d) for d in expr.shape)
                        for expr in exprs)
            _placement.split_reserve(total, device)
        for k, expr in enumerate(exprs):
            dev = _device_inputs(expr, E, device, seed=1000 * info.rank + k)
            for name in ("J", "R"):          # div and grad of one operator share J and D
                if name in dev:
                    dev[name] = shared.setdefault(name, dev[name])
            stages.append((expr, dev))
            out_dicts.append(measure.generate_out_arrays(q, expr, E, split=split))
        return stages, out_dicts

    def step_batch_of(op_):
        s_ = q.stream_ptr
        if len(op_.launches) == 1 and hasattr(op_.launches[0], "time_batch"):
            return lambda n: op_.launches[0].time_batch(n, s_)    # HIP events on the launch stream (fe_time_launches)
        return lambda n: op_.time_batch(n, s_)                    # same, through torch's event objects

    placement_report = {"mode": "separate", "what": "one torch allocation per array"}
    if args.placement == "split":
        from feinsum_amd import placement

        t_alloc = time.perf_counter()
        try:
            stages, out_dicts = separate_allocations(split=True)
        except Exception as exc:      # noqa: BLE001  (the allocator must not cost the line: torch allocations, and say so)
            print(f"bench.py: split allocator failed ({type(exc).__name__}: {str(exc)[:200]}); using torch allocations",
                  file=sys.stderr, flush=True)
            args.placement = "separate"
            stages, out_dicts = separate_allocations()
            placement_report = {"mode": "separate", "what": "one torch allocation per array",
                                "fallback": f"split allocator failed: {type(exc).__name__}: {str(exc)[:200]}"}
        sync()
        t_alloc = (time.perf_counter() - t_alloc) * 1e3
        if args.placement == "split":
            infos = [placement.split_info(t) for od in out_dicts for t in od.values()]
            pool = placement.split_stats(device)
            placement_report = {
                "mode": "split",
                "what": "one allocation per array; outputs from the split allocator (fe_split_alloc: 4 MiB pieces alternating between "
                        "two classes of physical memory, classified in groups of 128 MiB by a two-stream write probe; no arena, no "
                        "timing scan)",
                "output_pieces_by_class": [i.get("pieces_by_class", "torch allocation (below 8 MiB, or refused by the allocator: no second class found)") for i in infos],
                "output_bytes": sum(int(t.numel()) * t.element_size() for od in out_dicts for t in od.values()),
                "output_mapped_bytes": sum(i.get("mapped_bytes", 0) for i in infos),
                "allocator_ms": round(pool["setup_ms"] + pool["alloc_ms_total"], 3),
                "inputs_and_outputs_ready_ms": round(t_alloc, 1),
                "pool": {k: pool.get(k) for k in ("classes", "pieces_created", "groups_probed", "probes", "probe_ms", "spacers_created", "spacer_bytes_peak", "spacer_ms",
                                                    "search_ms", "search_ms_budget", "release_ms", "groups_discarded", "unsplit_arrays", "unsplit_refused", "pooled_bytes", "walk_gave_up")},
                # an output the allocator could not split (no second class of physical memory within its budget) is an ordinary torch
                # allocation (round 5; round 4 handed out an array of ONE class): not a split-placement result
                "degraded": bool(pool.get("unsplit_arrays", 0) or pool.get("unsplit_refused", 0)),
            }
    else:
        stages, out_dicts = separate_allocations()
    outs_all = [t for od in out_dicts for t in od.values()]
    op = bind(stages, out_dicts)
    prepared = any(getattr(b, "_prepared", None) for b in op._stages)

    s = q.stream_ptr
    # setup, untimed and not counted as warm-up steps (both figures are reported in the line): kernel attributes, first
    # touch of every page, device clocks and power state settled.  At least `--setup-launches` launches and at least
    # `--setup-seconds` of them: 105 launches (21 ms of work) into a fresh process the same bound launch still ran 3-4 %
    # slower than in the 2 s reference-protocol loop behind the timed region (0.1969-0.2010 against 0.1936-0.1943 ms in
    # four runs, 11 % at E = 1e5: profiles/r03/bench_grad_driver_style_short_setup.txt); round 2's arena scan had hidden
    # that behind its ~2000 launches
    t_setup = time.perf_counter()
    setup_launches = 0
    while setup_launches < args.setup_launches or time.perf_counter() - t_setup < args.setup_seconds:
        for _ in range(50):
            op.launch(s)
        setup_launches += 50
        sync()
    setup_seconds = time.perf_counter() - t_setup
    for _ in range(args.warmup):
        op.launch(s)

    step_batch = step_batch_of(op)

    def keep_busy():     # (a rank waiting in the opening barrier: ~1 ms of the same launches, untimed)
        for _ in range(max(1, min(50, int(1e-3 / max(setup_seconds / max(setup_launches, 1), 1e-6))))):
            op.launch(s)
        torch.cuda.current_stream(device).synchronize()     # (the launch stream only: the barrier's own kernel is still waiting on its stream)

    timed = timed_region(step_batch, args.steps, sync, device, keep_busy)
    wall_s, kernel_s = timed["wall"], timed["kernel"]
    launch_info = _hip.last_launch_info()      # what the launcher decided for the timed launches (fe_last_launch_info)
    pool_r = placement_report.get("pool") or {}
    per_rank = gather_rank_reports({"elements": E, "kernel_ms": timed["local_kernel"] / args.steps * 1e3, "wall_ms": timed["local_wall"] / args.steps * 1e3,
                                    "placement_mode": ("split" if placement_report.get("mode") == "split" else
                                                       "separate (split allocator failed)" if "fallback" in placement_report else "separate"),
                                    "unsplit_arrays": (pool_r.get("unsplit_arrays", 0) or 0) + (pool_r.get("unsplit_refused", 0) or 0), "allocator_ms": placement_report.get("allocator_ms", 0.0),
                                    "search_ms": pool_r.get("search_ms", 0.0) or 0.0, "release_ms": pool_r.get("release_ms", 0.0) or 0.0,
                                    "busy_calls": timed.get("opening_barrier_busy_calls", 0)}, device)

    # the reference's own protocol on the same bound launch (every rank, no barrier inside)
    protocol_ms = None
    if not args.no_protocol:
        dev_s, n_launch, host_s = 0.0, 0, 0.0
        for _ in range(measure.N_WARMUP_ROUNDS):
            op.launch(s)
        sync()
        # clocks / power / temperatures WHILE the launches run (the 2 s of this protocol; the K timed steps
        # above are over in milliseconds): rank 0's device, sampled from a side thread every 20 ms
        sampler = device_sampler(info.local_rank)
        with sampler:
            while n_launch < measure.N_MIN_TIMING_ROUNDS or host_s < measure.N_MIN_SIM_SECS:
                t0 = time.perf_counter()
                dev_s += step_batch(measure.LAUNCHES_PER_BATCH)      # fences like evt.wait()
                host_s += time.perf_counter() - t0
                n_launch += measure.LAUNCHES_PER_BATCH
        device_under_load = sampler.summary()
        protocol_ms = {"device": dev_s / n_launch * 1e3, "host": host_s / n_launch * 1e3, "launches": n_launch}

    # A/B outside the timed region: the same launch on one-torch-allocation-per-array operands
    separate_ms = None
    if args.placement != "separate" and not args.no_protocol:
        op_sep = bind(*separate_allocations(), variant_=args.variant)
        sb = step_batch_of(op_sep)
        sb(max(args.warmup, 10))
        separate_ms = sb(args.steps) / args.steps * 1e3
        del op_sep, sb

    # A/B outside the timed region: the same launch on the same operands with the STATIC walk (the timed launches hand their
    # tiles to the waves by tickets behind two static rounds: feinsum_amd/csrc/fe_common.h, dynamic walk)
    static_walk_ms = None
    rounds_setting = _hip.set_tail_rounds(-1)
    try:
        if not args.no_protocol:
            step_batch(max(args.warmup, 10))
            static_walk_ms = step_batch(args.steps) / args.steps * 1e3
    finally:
        _hip.set_tail_rounds(rounds_setting)
    dyn = bool(launch_info.get("dynamic_walk"))
    walk_report = {"mode": "tickets behind the static rounds" if dyn else "static",
                   "tiles": launch_info.get("tiles"), "static_tiles": launch_info.get("static_tiles"),
                   "grid": [launch_info.get("blocks"), launch_info.get("waves_per_block")], "kernel": launch_info.get("kind"),
                   # (short launches: the ragged last round as quarter tiles; div's B build inside its matrix phase -- fe_last_launch_info)
                   "quarter_tile_tail": bool(launch_info.get("quarter_tail")), "staggered_start": bool(launch_info.get("staggered_start")), "interleaved_b_build": bool(launch_info.get("interleaved")),
                   "rule": "tickets in launches of four and a half or more rounds (fe_set_tail_rounds: %s)" % ("all" if rounds_setting >= (1 << 20) else rounds_setting),
                   # (the A/B is a different launch only when the timed one walked dynamically)
                   "kernel_ms_static_walk": None if (static_walk_ms is None or not dyn) else round(static_walk_ms, 5)}

    want_gather = parallel.in_group() and args.gather_fields == "on"
    # A/B outside the timed region: the streamed operand fetched the other way (plain loads while the launch's inputs fit the
    # Infinity Cache, non-temporal otherwise: feinsum_amd/csrc/fe_common.h, fe_set_temporal_loads_mib)
    in_bytes = sum(int(t.numel()) * t.element_size() for t in {id(t): t for _, d in stages for t in d.values()}.values())
    mib_setting = _hip.set_temporal_loads_mib(0)
    plain = bool(launch_info.get("temporal_loads"))       # (the launcher's own decision, not a re-derivation of its rule)
    loads_report = {"threshold_mib": mib_setting, "threshold_note": "grad; x 280/248 div, x 310/248 div + grad, x 320/248 launches with face-mass in them (fe_set_temporal_loads_mib)",
                    "launch_input_mib": round(in_bytes / 2**20, 1),
                    "streamed_operand": "plain loads (what the 256 MiB Infinity Cache holds of the launch's inputs a repeated launch finds there)"
                    if plain else "non-temporal loads"}
    try:
        if not args.no_protocol and plain:
            step_batch(max(args.warmup, 10))
            if not _hip.last_launch_info().get("temporal_loads"):     # the A/B launch really ran the other way
                loads_report["kernel_ms_non_temporal_loads"] = round(step_batch(args.steps) / args.steps * 1e3, 5)
    finally:
        _hip.set_temporal_loads_mib(mib_setting)

    # A/B outside the timed region: the output stores the other way (write-through in short grad launches -- static walk, outputs
    # of at most fe_set_write_through_mib MiB -- non-temporal otherwise: feinsum_amd/csrc/fe_common.h)
    out_bytes = sum(int(t.numel()) * t.element_size() for t in outs_all)
    wt_setting = _hip.set_write_through_mib(0)
    write_through = bool(launch_info.get("write_through_stores"))
    stores_report = {"threshold_mib": wt_setting, "launch_output_mib": round(out_bytes / 2**20, 1),
                     "policy": "write-through (a short launch: nothing dirty left in the L2s at its end)" if write_through else "non-temporal"}
    try:
        if not args.no_protocol and write_through:
            step_batch(max(args.warmup, 10))
            if not _hip.last_launch_info().get("write_through_stores"):
                stores_report["kernel_ms_non_temporal_stores"] = round(step_batch(args.steps) / args.steps * 1e3, 5)
    finally:
        _hip.set_write_through_mib(wt_setting)

    total, reduction_ms, allgather_ms = exchange_results(outs_all, sync)
    finite = bool(torch.isfinite(total).all().item()) and bool((total[:, 1] > 0).all().item())

    # flops of the whole job = sum over ranks (ranks may hold different element counts)
    flops_all = flops_step
    ranks_seen = 1
    if parallel.in_group():
        import torch.distributed as dist

        ranks_seen = dist.get_world_size()

        t = torch.tensor([flops_step], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t)
        flops_all = float(t.item())

    if info.rank == 0:
        extra = {"setup_seconds": round(setup_seconds, 3),
                 "result_reduction_ms": round(reduction_ms, 3), "result_allgather_ms": round(allgather_ms, 3),
                 "result_finite": finite, "kernel_source_sha": kernel_source_sha(),
                 "operator_prepared": prepared, "placement": placement_report,
                 "kernel_ms_separate_allocations": None if separate_ms is None else round(separate_ms, 5),
                 "walk": walk_report, "loads": loads_report, "stores": stores_report,
                 # `value` / `ms_per_step`: MAX over ranks of each rank's own time to its own synchronize (no data-path
                 # collective to wait for); the same region with the closing barrier + synchronize inside, for comparison:
                 "ms_per_step_barrier_inclusive": round(timed["wall_barrier"] / args.steps * 1e3, 5),
                 "per_rank": per_rank,
                 "dist_backend": info.backend if parallel.in_group() else None,
                 # the optional full-field exchange runs BEHIND this line (rank 0 reports it on stderr): whatever happens
                 # inside a 6.7 GB-per-GPU collective cannot cost the scaling record
                 "field_allgather": "off" if not want_gather else "runs after this line; reported on stderr as `field_allgather {...}`",
                 # how many ranks the process group really holds (1 without a group), and who started them
                 "ranks_seen": ranks_seen,
                 "launcher": os.environ.get("FEINSUM_BENCH_LAUNCHER") or
                             ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                              "external" if "WORLD_SIZE" in os.environ else "single process")}
        if protocol_ms is not None:
            extra["protocol_ms_per_step"] = round(protocol_ms["device"], 5)
            extra["protocol"] = {"what": "reference timing protocol (src/feinsum/measure.py:248-275): 5 warm-ups, "
                                         "batches of 5 launches, >= 10 launches and >= 2 s; HIP events per batch",
                                 "launches": protocol_ms["launches"],
                                 "host_ms_per_step": round(protocol_ms["host"], 5)}
            extra["device_under_load"] = device_under_load
        line = compose_line(workload=args.workload, n_gpus=info.world_size, steps=args.steps, warmup=args.warmup,
                            setup_launches=setup_launches,
                            wall_s=wall_s, kernel_s=kernel_s, flops_step_all=flops_all, flops_step_rank0=flops_step,
                            bytes_step_rank0=bytes_step, elems_rank0=E, elems_total=args.elems_total,
                            variant=variant, device_name=q.device.name, entry_points=op.entry_points,
                            extra=extra)
        rec, note = committed_counters(args.workload, E)
        if rec is not None and not args.no_fuse and not args.prepare:      # (the profiled kernels are the default ones)
            line["roofline"]["traffic"] = rec.get("hbm_bytes_per_launch")
            line["roofline"]["traffic_source"] = {k: rec.get(k) for k in ("kernel", "source_sha", "profile", "E")}
            line["mfma_util"] = rec.get("mfma_util")
            line["mfma_util_source"] = rec.get("mfma_util_formula")
            # the PMC passes run with one allocation per array (their launch time is the placement lottery of
            # DESIGN section 3d); the busy cycles per launch are fixed, so the same ratio at THIS run's launch time:
            busy = ((rec.get("counters") or {}).get("SQ_VALU_MFMA_BUSY_CYCLES") or {}).get("mean")
            sclk = (((extra.get("device_under_load") or {}).get("freq_sclk_mhz")) or {}).get("mean")
            if busy and sclk:
                line["mfma_busy_cycles_per_launch"] = busy
                line["mfma_util_at_this_launch_time"] = round(busy / (1024 * (kernel_s / args.steps) * sclk * 1e6), 4)
        else:
            line["roofline"]["traffic_note"] = note or "the committed PMC profile is of the default (fused, unprepared) launch"
            line["mfma_util"] = None
        if info.world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, full=args.cpu_baseline == "full")
        print(json.dumps(line), flush=True)

    # the optional full-field exchange (off the clock; never part of `value`): the one place xGMI carries data.  It runs
    # AFTER rank 0 has printed the line.  No rank enters a collective alone: the shard lengths are computed alike on every
    # rank, every rank stages its receive buffers first (the likely failure: 6.7 GB per GPU for grad at 8 ranks) and the
    # ranks agree on the outcome with one all-reduce of a flag (parallel.allgather_fields_timed).
    if want_gather:
        lengths = {E} if not args.elems_total else \
            {b - a for a, b in (parallel.shard_bounds(args.elems_total, info.world_size, r) for r in range(info.world_size))}
        if len(lengths) != 1:
            field_gather = {"skipped": "shards of unequal length"}
        else:
            try:
                axes = [[i for i, d in enumerate(expr.shape) if isinstance(d, f.SizeParam)][0]
                        for expr in exprs for _ in expr.output_names]
                g = parallel.allgather_fields_timed(list(zip(outs_all, axes)), sync)
                if "sums" in g:
                    sums = g.pop("sums")
                    g["matches_reduction"] = bool(all(abs(a - float(b)) <= 1e-9 * max(1.0, abs(float(b)))
                                                      for a, b in zip(sums, total[:, 0].tolist())))
                    g["ms"], g["gbps_per_gpu"] = round(g["ms"], 3), round(g["gbps_per_gpu"], 1)
                    g["backend"] = info.backend           # nccl = RCCL over xGMI; gloo (rehearsals) goes through host memory
                    g["xgmi_peak_gbps_per_gpu"] = 7 * 153.0
                field_gather = g
            except Exception as exc:      # noqa: BLE001  (reported; the line is out already)
                field_gather = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if info.rank == 0:
            print("field_allgather " + json.dumps(field_gather), file=sys.stderr, flush=True)

    if parallel.in_group():
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
