#!/bin/bash
# HBM bytes per launch of the float32 kernels (FETCH_SIZE / WRITE_SIZE, one pass each; gfx950: FETCH_SIZE counts 32-byte
# units, WRITE_SIZE KiB: MI355X_MICROARCH.md):  bash tools/f32_pmc.sh [E]   -> gpurun_out/f32pmc/summary.txt
set -e
E=${1:-1000000}
keep=$PWD/gpurun_out/f32pmc; repo=$PWD; out=/tmp/f32pmc
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
for w in grad div face_mass; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/${w}_$c" -o pmc -- python3 "$repo/tools/f32_pmc_run.py" $w $E > /dev/null 2> "$out/${w}_$c.err" \
      || { tail -5 "$out/${w}_$c.err"; exit 1; }
  done
done
python3 - "$out" $E > "$keep/summary.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, E = sys.argv[1], int(sys.argv[2])
alg = {"grad": 596, "div": 596, "face_mass": 1536}
for w in ("grad", "div", "face_mass"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = defaultdict(float)
        for path in glob.glob(f"{out}/{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == c and "f32" in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        v = sorted(per.values())
        tot[c] = v[len(v) // 2] if v else float("nan")
        n = len(v)
    fetch, write = tot["FETCH_SIZE"] * 32, tot["WRITE_SIZE"] * 1024
    print(f"{w} float32 E={E}: fetch {fetch/1e6:.1f} MB  write {write/1e6:.1f} MB  total {(fetch+write)/1e6:.1f} MB  algorithmic {alg[w]*E/1e6:.1f} MB  ratio {(fetch+write)/(alg[w]*E):.4f}  ({n} launches)")
PY
cat "$keep/summary.txt"
