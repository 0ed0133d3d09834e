#!/bin/bash
# bench.py --workload pipeline on the experiment build with body orders 0 (every block div, grad, lift) and 3 (the younger half
# grad first), static walk and tickets; one process each:   bash tools/build_experiments.sh && bash tools/pipeline_order_ab.sh
export FEINSUM_HIP_LIB=$PWD/build/libfeinsum_hip_exp.so
for rep in 1 2; do for rounds in -1 1048576; do for order in 0 3; do
  FE_FUSED_ORDER=$order FEINSUM_TAIL_ROUNDS=$rounds python3 bench.py --workload pipeline --no-cpu-baseline --no-protocol --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('pipeline rounds=$rounds order=$order: kernel %.4f ms  frac %.4f' % (d.get('kernel_ms', float('nan')), d['roofline']['frac']), flush=True)
"
done; done; done
