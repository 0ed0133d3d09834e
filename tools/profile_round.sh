#!/bin/bash
# Kernel-trace summaries and HBM counters of the bench workloads (run on the GPU box):
#   bash tools/profile_round.sh r01b
# writes gpurun_out/prof_<tag>/...; copy the *_kernel_stats.csv / *_pmc.csv into profiles/<round>/.
set -e
tag=${1:-r01b}
keep=$PWD/gpurun_out/prof_$tag
repo=$PWD
out=/tmp/prof_$tag          # raw rocprofv3 output is large; only the summaries are kept
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
for w in grad div facemass graddiv pipeline; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w" -o $w -- python3 "$repo/bench.py" --workload $w --no-cpu-baseline \
      > "$out/bench_${w}_under_rocprof.json" 2> "$out/$w.err" || { tail -5 "$out/$w.err"; exit 1; }
done
# HBM traffic of the fused / batched kernels: separate --pmc passes (MI355X_MICROARCH.md, HBM section)
for w in graddiv pipeline; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_${w}_$c" -o pmc -- python3 "$repo/bench.py" --workload $w \
        --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> "$out/pmc_${w}_$c.err" || { tail -5 "$out/pmc_${w}_$c.err"; exit 1; }
  done
done
for w in grad div facemass graddiv pipeline; do
  cp "$out/bench_${w}_under_rocprof.json" "$keep/"
  f=$(find "$out/$w" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$keep/${w}_kernel_stats.csv"
done
for w in graddiv pipeline; do
  for c in FETCH_SIZE WRITE_SIZE; do
    f=$(find "$out/pmc_${w}_$c" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$keep/pmc_${w}_$c.txt" <<'PY'
import csv, sys, collections
tot, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    tot[k] += float(r["Counter_Value"]); n[k] += 1
open(sys.argv[2], "w").write("".join(f"{k[0]} {k[1]} mean_per_dispatch={tot[k]/n[k]:.1f} dispatches={n[k]}\n" for k in sorted(tot)))
PY
  done
done
ls -la "$keep"; du -sh "$out"
