#!/bin/bash
# The N > 1 paths of bench.py on ONE GPU (the ranks share it through gloo; the pool allows six GPU processes at once):
#   bash tools/rehearse_ranks.sh [outdir]
# six self-spawned ranks (six allocator pools searching one device at the same instant, each bounded by the pool-wide deadline),
# and torch.distributed.run with four ranks on the pipeline with the field gather behind the line.
out=${1:-gpurun_out/rehearse}; mkdir -p $out
FEINSUM_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu-baseline > $out/selfspawn6.json 2> $out/selfspawn6.err; tail -2 $out/selfspawn6.err
FEINSUM_DIST_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --steps 20 --warmup 5 --workload pipeline --elems-per-gpu 250000 --gather-fields on --no-cpu-baseline > $out/torchrun4_pipeline_gather.json 2> $out/torchrun4.err; grep "field_allgather" $out/torchrun4.err | cut -c1-300
python3 - $out <<'PY'
import json, sys
for name in ("selfspawn6.json", "torchrun4_pipeline_gather.json"):
    for l in open(f"{sys.argv[1]}/{name}"):
        if l.startswith("{"):
            d = json.loads(l)
            p = d.get("placement") or {}
            print(name, "n_gpus", d["n_gpus"], "ranks_seen", d["ranks_seen"], d["launcher"], "value", round(d["value"]), "ms_per_step", d["ms_per_step"],
                  "barrier-inclusive", d.get("ms_per_step_barrier_inclusive"), "finite", d["result_finite"], "degraded", p.get("degraded"))
            for r in d.get("per_rank", []):
                print("   ", r)
PY
