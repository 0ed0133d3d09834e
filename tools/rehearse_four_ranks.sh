#!/bin/bash
# Round 4: the N > 1 paths of bench.py with FOUR ranks sharing the one GPU through gloo -- four allocator pools searching the
# same device at the same instant (each bounded to 2.5 s by bench.py), the field gather behind the line, torch.distributed.run.
out=${1:-gpurun_out/rehearse4}; mkdir -p $out
FEINSUM_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > $out/selfspawn4.json 2> $out/selfspawn4.err; tail -2 $out/selfspawn4.err
FEINSUM_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --steps 20 --warmup 5 --workload pipeline --elems-per-gpu 250000 --gather-fields on --no-cpu-baseline > $out/torchrun4_pipeline_gather.json 2> $out/torchrun4.err; grep "field_allgather" $out/torchrun4.err | cut -c1-400
python3 - $out <<'PY'
import json, sys
for name in ("selfspawn4.json", "torchrun4_pipeline_gather.json"):
    for l in open(f"{sys.argv[1]}/{name}"):
        if l.startswith("{"):
            d = json.loads(l)
            p = d.get("placement") or {}
            print(name, "n_gpus", d["n_gpus"], "ranks_seen", d["ranks_seen"], d["launcher"], "value", round(d["value"]), "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"],
                  "finite", d["result_finite"], "allocator_ms", p.get("allocator_ms"), "unsplit", (p.get("pool") or {}).get("unsplit_arrays"), "field_allgather:", d.get("field_allgather"))
PY
