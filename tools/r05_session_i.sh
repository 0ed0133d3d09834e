#!/bin/bash
out=gpurun_out/r05i; mkdir -p $out
timeout -k 10 120 build/store_footprint > $out/store_footprint.txt 2>&1; cat $out/store_footprint.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleaved or families_vs_oracle" > $out/pytest_parity_subset.log 2>&1; tail -3 $out/pytest_parity_subset.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for w in grad div graddiv pipeline; do timeout -k 10 300 python3 bench.py --workload $w --elems-per-gpu 100000 --no-cpu-baseline > $out/bench_${w}_1e5.json 2>> $out/bench.err; done
python3 - $out <<'PY'
import json, sys, glob
for fn in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(fn):
        if l.startswith("{"):
            d = json.loads(l); w = d["walk"]
            print(fn.split("/")[-1], "kernel_ms", d["kernel_ms"], "frac", d["roofline"]["frac"], "| walk", w["mode"], w.get("kernel"), "static A/B", w["kernel_ms_static_walk"], "| separate", d.get("kernel_ms_separate_allocations"), "| protocol", d.get("protocol_ms_per_step"))
PY
