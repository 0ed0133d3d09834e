#!/bin/bash
# bench.py lines of one workload with the dynamic tail off (-1) and with 1 ... 12 dynamic rounds (FEINSUM_TAIL_ROUNDS), one fresh
# process each:   bash tools/tail_rounds_ab.sh grad 1000000 "-1 1 3 6 12" > gpurun_out/.../tail.txt
w=${1:-grad}; E=${2:-1000000}; rounds=${3:--1 1 3 6 12}
for r in $rounds; do
  FEINSUM_TAIL_ROUNDS=$r python3 bench.py --workload $w --elems-per-gpu $E --no-cpu-baseline --no-protocol --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print('$w E=$E rounds=$r: kernel %.4f ms  frac %.4f  ms_per_step %.4f' % (d.get('kernel_ms', float('nan')), d['roofline']['frac'], d['ms_per_step']), flush=True)
"
done
