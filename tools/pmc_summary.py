#!/usr/bin/env python
"""
Summarise rocprofv3 --pmc passes of one bench workload into profiles/traffic_<workload>.json.

    python tools/pmc_summary.py <workload> <E> <out.json> <pass_dir> [<pass_dir> ...]

Every <pass_dir> is the -d directory of one `rocprofv3 --pmc ... --kernel-trace --output-format
csv -- python3 bench.py ...` run (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Per counter the mean over the dispatches of the
workload's dominant kernel (the one with the largest summed duration) is taken, then:

  hbm_read_bytes_per_launch  = 2 x FETCH_SIZE x 1024   (gfx950: FETCH_SIZE tallies 128-B requests
                                                        as 64 B -- MI355X_MICROARCH.md, HBM)
  hbm_write_bytes_per_launch = WRITE_SIZE x 1024        (exact for 16-B-per-lane streaming stores)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)
              (rocprofiler-sdk counter_defs.yaml `MfmaUtil`: busy cycles summed over SIMDs /
              (active cycles x SIMD count); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs)

The record carries the kernel name and the hash of the kernel sources it was taken on
(bench.kernel_source_sha): bench.py prints these numbers only while that hash matches.
"""

from __future__ import annotations

import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def read_pass(d: Path):
    """{kernel: {counter: [values per dispatch]}}, {kernel: [durations ns]}"""
    vals: dict = defaultdict(lambda: defaultdict(list))
    durs: dict = defaultdict(list)
    for f in d.rglob("*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"],)
            if key not in seen:
                seen.add(key)
                durs[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return vals, durs


def main() -> None:
    workload, E, out = sys.argv[1], int(sys.argv[2]), Path(sys.argv[3])
    merged: dict = defaultdict(dict)
    dur_sum: dict = defaultdict(float)
    ndisp: dict = {}
    for d in sys.argv[4:]:
        vals, durs = read_pass(Path(d))
        for k, cs in vals.items():
            for c, v in cs.items():
                merged[k][c] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
        for k, v in durs.items():
            dur_sum[k] += sum(v)
            ndisp[k] = len(v)
    if not dur_sum:
        raise SystemExit("no counter_collection.csv found")
    # the workload's kernel: the library's own (fe::...), not the allocator's classification probe and not torch's fills
    ours = {k: v for k, v in dur_sum.items() if "fe::" in k and "split_probe_kernel" not in k} or dur_sum
    kernel = max(ours, key=ours.get)
    c = merged[kernel]
    mean = lambda name: c[name]["mean"] if name in c else None   # noqa: E731

    import bench

    rec = {"workload": workload, "E": E, "kernel": kernel, "source_sha": bench.kernel_source_sha(),
           "profile": "rocprofv3 --pmc <one counter group per pass> --kernel-trace -- python3 bench.py "
                      f"--workload {workload} --no-cpu-baseline --no-protocol --setup-launches 3 --setup-seconds 0 --steps 10 --warmup 2 --elems-per-gpu {E}, default placement "
                      "(outputs from the split allocator; tools/profile_round.sh)",
           "dispatches_per_pass": ndisp[kernel], "counters": c}
    if mean("FETCH_SIZE") is not None:
        rec["hbm_read_bytes_per_launch"] = 2.0 * mean("FETCH_SIZE") * 1024.0
    if mean("WRITE_SIZE") is not None:
        rec["hbm_write_bytes_per_launch"] = mean("WRITE_SIZE") * 1024.0
    if "hbm_read_bytes_per_launch" in rec and "hbm_write_bytes_per_launch" in rec:
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
    rec["correction"] = ("gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM) -> read bytes = "
                         "2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 exact for 16-B/lane streaming stores")
    if mean("SQ_VALU_MFMA_BUSY_CYCLES") is not None and mean("GRBM_GUI_ACTIVE"):
        rec["mfma_util"] = round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / (mean("GRBM_GUI_ACTIVE") / 8.0 * 1024.0), 4)
        rec["mfma_util_formula"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), committed "
                                    "rocprofv3 --pmc pass (profiles/traffic_*.json)")
    if mean("SQ_VALU_MFMA_BUSY_CYCLES") is not None and mean("SQ_BUSY_CYCLES"):
        # second estimate of the same ratio: SQ_BUSY_CYCLES is reported summed over the 32 shader engines
        rec["mfma_util_by_sq_busy"] = round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / (mean("SQ_BUSY_CYCLES") / 32.0 * 1024.0), 4)
    out.write_text(json.dumps(rec, indent=1, sort_keys=True) + "\n")
    brief = {k: rec.get(k) for k in ("kernel", "hbm_bytes_per_launch", "mfma_util", "source_sha")}
    print(json.dumps(brief))


if __name__ == "__main__":
    main()
