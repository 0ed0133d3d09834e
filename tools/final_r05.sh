#!/bin/bash
# Round-5 final measurements in one GPU call: outputs under gpurun_out/final_r05/ (copied to profiles/r05/ afterwards).
out=gpurun_out/final_r05; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_grad_driver$i.json 2>> $out/bench.err; done
for w in div facemass graddiv pipeline; do python3 bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2>> $out/bench.err; done
for w in grad div facemass graddiv pipeline; do python3 bench.py --workload $w --elems-per-gpu 100000 --no-cpu-baseline > $out/bench_${w}_1e5.json 2>> $out/bench.err; done
for w in grad div pipeline; do python3 bench.py --workload $w --elems-per-gpu 200000 --no-cpu-baseline > $out/bench_${w}_2e5.json 2>> $out/bench.err; done
python3 bench.py --elems-per-gpu 8000000 --no-cpu-baseline > $out/bench_grad_8e6.json 2>> $out/bench.err
python3 bench.py --workload pipeline --elems-per-gpu 8000000 --no-cpu-baseline > $out/bench_pipeline_8e6.json 2>> $out/bench.err
for f in $out/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        w = d.get("walk") or {}
        print(sys.argv[1].split("/")[-1], "value %.0f" % d["value"], "ms_per_step %.5f" % d["ms_per_step"], "kernel_ms", d.get("kernel_ms"), "frac", d["roofline"]["frac"],
              "| walk", w.get("mode", "")[:7], w.get("kernel"), "static A/B", w.get("kernel_ms_static_walk"), "| separate", d.get("kernel_ms_separate_allocations"),
              "| nt loads", (d.get("loads") or {}).get("kernel_ms_non_temporal_loads"), "| traffic", d["roofline"].get("traffic"), "mfma_util", d.get("mfma_util"))
PY
done | tee $out/summary.txt
