#!/bin/bash
# The N > 1 paths of bench.py on ONE GPU (two ranks share it through gloo): the self-spawning launcher and torch.distributed.run.
out=${1:-gpurun_out/rehearse2}; mkdir -p $out
FEINSUM_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $out/selfspawn2.json 2> $out/selfspawn2.err; tail -2 $out/selfspawn2.err
FEINSUM_DIST_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --workload pipeline --no-cpu-baseline > $out/torchrun2_pipeline.json 2> $out/torchrun2.err; tail -2 $out/torchrun2.err
python3 - $out <<'PY'
import json, sys
for name in ("selfspawn2.json", "torchrun2_pipeline.json"):
    for l in open(f"{sys.argv[1]}/{name}"):
        if l.startswith("{"):
            d = json.loads(l)
            print(name, "n_gpus", d["n_gpus"], "ranks_seen", d["ranks_seen"], d["launcher"], "value", round(d["value"]), "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "finite", d["result_finite"])
PY
