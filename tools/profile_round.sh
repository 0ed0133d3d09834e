#!/bin/bash
# Kernel-trace summaries and PMC counters of the bench workloads, on the library as built from the
# current sources (run on the GPU box):
#   bash tools/profile_round.sh r03 [workloads...]
#   ELEMS=100000 bash tools/profile_round.sh r05 grad div     (another element count: files get the suffix _E100000)
# writes gpurun_out/prof_<tag>/: <w>_kernel_stats.csv, bench_<w>_under_rocprof.json, traffic_<w>.json
# (copy the traffic files to profiles/ and the rest to profiles/<round>/).
# Under rocprofv3 the program goes directly after `--` (python3, no wrapper); counters are collected
# in their own passes with --kernel-trace only.  All passes run the default placement (outputs from the split allocator),
# i.e. the layout of the bench line -- round 2's PMC passes ran on torch allocations, a placement lottery.
set -e
tag=${1:-r03}; shift || true
workloads=${@:-grad div facemass graddiv pipeline}
E=${ELEMS:-1000000}
sfx=""; [ "$E" != 1000000 ] && sfx="_E$E"
steps=3000; [ "$E" -lt 500000 ] && steps=10000
keep=$PWD/gpurun_out/prof_$tag
repo=$PWD
out=/tmp/prof_$tag          # raw rocprofv3 output is large; only the summaries are kept
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for w in $workloads; do
  echo "== $w: kernel trace"
  # --no-protocol: no 2 s reference-protocol loop and no A/B launch on torch allocations in the profiled process, so that
  # every dispatch of the workload's kernel is the SAME bound launch on the SAME (split allocator) arrays: the average of
  # <w>_kernel_stats.csv is the timed kernel alone (round 2's average mixed in ~200 positions of an arena scan); 3000 timed
  # steps so that the 120 setup / warm-up launches (first touch, clocks ramping: up to 280 us) weigh < 1 % in the average
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w" -o $w -- python3 "$repo/bench.py" --workload $w --no-cpu-baseline \
      --no-protocol --steps $steps --elems-per-gpu $E > "$out/bench_${w}${sfx}_under_rocprof.json" 2> "$out/$w.err" || { tail -5 "$out/$w.err"; exit 1; }
  cp "$out/bench_${w}${sfx}_under_rocprof.json" "$keep/"
  f=$(find "$out/$w" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$keep/${w}${sfx}_kernel_stats.csv"
  n=0
  for group in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    echo "== $w: pmc pass $n ($group)"
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/pmc_${w}_$n" -o pmc -- python3 "$repo/bench.py" --workload $w \
        --no-cpu-baseline --no-protocol --setup-launches 3 --setup-seconds 0 --steps 10 --warmup 2 --elems-per-gpu $E > /dev/null 2> "$out/pmc_${w}_$n.err" \
        || { tail -5 "$out/pmc_${w}_$n.err"; exit 1; }
  done
  python3 "$repo/tools/pmc_summary.py" $w $E "$keep/traffic_$w$sfx.json" "$out/pmc_${w}_1" "$out/pmc_${w}_2" "$out/pmc_${w}_3"
done
ls -la "$keep"; du -sh "$out"
