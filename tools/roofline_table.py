#!/usr/bin/env python
"""The table of DESIGN.md section 4 from the committed bench lines:  python tools/roofline_table.py [profiles/r05]"""
import glob
import json
import sys
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "profiles/r05")


def lines(pattern):
    out = []
    for f in sorted(glob.glob(str(root / pattern))):
        for line in open(f):
            if line.startswith("{"):
                out.append((Path(f).name, json.loads(line)))
    return out


def rng(vals, fmt):
    vals = [v for v in vals if v is not None]
    if not vals:
        return "—"
    lo, hi = min(vals), max(vals)
    return fmt % lo if fmt % lo == fmt % hi else f"{fmt % lo}–{fmt % hi}"


rows = [("grad, E = 10⁶ (the driver's command)", "bench_grad_driver*.json"), ("div", "bench_div.json"), ("face-mass × 4", "bench_facemass.json"),
        ("div + grad, one launch", "bench_graddiv.json"), ("div + grad + face-mass × 4, one launch", "bench_pipeline.json"),
        ("grad, 8·10⁶", "bench_grad_8e6.json"), ("pipeline, 8·10⁶", "bench_pipeline_8e6.json"),
        ("grad, 2·10⁵", "bench_grad_2e5.json"), ("div, 2·10⁵", "bench_div_2e5.json"), ("pipeline, 2·10⁵", "bench_pipeline_2e5.json"),
        ("grad, 10⁵", "bench_grad_1e5*.json"), ("div, 10⁵", "bench_div_1e5*.json"), ("face-mass × 4, 10⁵", "bench_facemass_1e5*.json"),
        ("div + grad, 10⁵", "bench_graddiv_1e5*.json"), ("pipeline, 10⁵", "bench_pipeline_1e5*.json")]
print("| launch | bytes per launch | kernel time (HIP events, K timed steps) | of the roofline | whole-job GFLOP/s (wall clock) | same process: static walk / torch arrays | PMC: HBM bytes ÷ algorithmic, matrix pipes busy |")
print("|---|---|---|---|---|---|---|")
for name, pat in rows:
    ds = [d for _, d in lines(pat)]
    if not ds:
        continue
    unit = 1e3 if ds[0]["kernel_ms"] < 0.15 else 1.0
    u = "µs" if unit == 1e3 else "ms"
    fmt = "%.1f" if unit == 1e3 else ("%.4f" if ds[0]["kernel_ms"] < 1 else "%.3f")
    alg = ds[0]["roofline"]["achieved"] * ds[0]["kernel_ms"] * 1e6          # GB/s x ms -> bytes
    traffic = [d["roofline"].get("traffic") for d in ds]
    ratio = rng([t / alg if t else None for t in traffic], "%.4f")
    static = rng([(d.get("walk") or {}).get("kernel_ms_static_walk") and (d["walk"]["kernel_ms_static_walk"] * unit) for d in ds], fmt)
    sep = rng([d.get("kernel_ms_separate_allocations") and d["kernel_ms_separate_allocations"] * unit for d in ds], fmt)
    print(f"| {name} | {alg / 1e9:.3f} GB | {rng([d['kernel_ms'] * unit for d in ds], fmt)} {u} | **{rng([d['roofline']['frac'] for d in ds], '%.3f')}** | "
          f"{rng([d['value'] for d in ds], '%.0f')} | {static} / {sep} {u} | {ratio} ×, {rng([d.get('mfma_util') for d in ds], '%.3f')} |")
