#!/bin/bash
# Round 5, GPU session E: stamps of the interleaved div kernel; tickets in three-round launches again (with write-through stores)
out=gpurun_out/r05e; mkdir -p $out
for E in 100000 98304; do
  FE_DIV_ILV=1 FE_DUMP_STAMPS=$out/stamps_divilv_$E.csv timeout -k 10 120 build/fe_check_exp ab div $E 5 50 0,1128 > $out/stamps_divilv_$E.txt 2>&1; tail -3 $out/stamps_divilv_$E.txt
  python3 tools/tile_stamps_report.py $out/stamps_divilv_$E.csv.tiles.csv div > $out/tiles_divilv_$E.txt 2>&1
done
for w in grad div pipeline; do
  for E in 100000 131072 160000; do
    FEINSUM_TAIL_MIN_ROUNDS=3 timeout -k 10 300 python3 bench.py --workload $w --elems-per-gpu $E --no-cpu-baseline > $out/bench_${w}_${E}_tickets3.json 2>> $out/bench.err
    python3 - $out/bench_${w}_${E}_tickets3.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); w = d["walk"]
        print(sys.argv[1].split("/")[-1], "kernel_ms", d["kernel_ms"], "frac", d["roofline"]["frac"], "| walk", w["mode"], "static A/B", w["kernel_ms_static_walk"], "| stores", d["stores"]["policy"][:14], d["stores"].get("kernel_ms_non_temporal_stores"), "| separate", d.get("kernel_ms_separate_allocations"))
PY
  done
done
