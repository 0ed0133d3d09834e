#!/bin/bash
# Round 5, GPU session B: per-tile stamps at E = 1e5 / 98304 (grad, div), the new ticket-ownership tests and the whole GPU suite,
# a six-rank rehearsal of bench.py on the one GPU (gloo; the pool allows six GPU processes at once)
out=gpurun_out/r05b; mkdir -p $out
for fam in grad div; do
  v=1032; [ $fam = div ] && v=1128
  for E in 100000 98304 200000; do
    FE_DUMP_STAMPS=$out/stamps_${fam}_$E.csv timeout -k 10 120 build/fe_check_exp ab $fam $E 5 50 0,$v > $out/stamps_${fam}_$E.txt 2>&1
    tail -3 $out/stamps_${fam}_$E.txt
    python3 tools/tile_stamps_report.py $out/stamps_${fam}_$E.csv.tiles.csv $fam > $out/tiles_${fam}_$E.txt 2>&1
  done
done
timeout -k 10 600 python -m pytest tests/test_gpu_streams.py -m gpu -x -q > $out/pytest_streams.log 2>&1; tail -3 $out/pytest_streams.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
FEINSUM_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu-baseline > $out/selfspawn6.json 2> $out/selfspawn6.err; tail -3 $out/selfspawn6.err
python3 - $out/selfspawn6.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("n_gpus", d["n_gpus"], "value", d["value"], "ms_per_step", d["ms_per_step"], "barrier-inclusive", d.get("ms_per_step_barrier_inclusive"), "placement", json.dumps(d["placement"].get("pool")), "degraded", d["placement"].get("degraded"))
        for r in d.get("per_rank", []): print("  ", r)
PY
