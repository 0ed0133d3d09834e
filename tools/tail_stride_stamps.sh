#!/bin/bash
# Per-pool end times of the dynamic walk (grad, E = 1e6, hipMalloc arrays) for several spacings of the ticket counters: rebuilds
# the experiment library on the box per spacing and reads the per-wave time stamps.   bash tools/tail_stride_stamps.sh "2176 2624 16512"
set -e
cp feinsum_amd/csrc/fe_common.h /tmp/fe_common.h.keep
for stride in ${1:-2176 2624 16512}; do
  sed -i "s/^constexpr int kTailStride = [0-9]*;/constexpr int kTailStride = $stride;/" feinsum_amd/csrc/fe_common.h
  bash tools/build_experiments.sh > /dev/null 2>&1
  for rep in 1 2; do
    FE_DUMP_STAMPS=/tmp/stamps.csv ./build/fe_check_exp ab grad 1000000 5 20 0,1032 2>&1 | grep "variant      0\|loop end" | sed "s/^/stride $stride: /"
    python3 - <<'PY'
import csv, numpy as np
rows = list(csv.DictReader(open("/tmp/stamps.csv")))
end = np.array([float(r["loop_end_us"]) for r in rows]); w = np.arange(len(rows)); pool = ((w // 4) >> 3) & 15
print("    per pool mean end:", " ".join("%.0f" % end[pool == p].mean() for p in range(16)), " spread %.1f" % (max(end[pool == p].mean() for p in range(16)) - min(end[pool == p].mean() for p in range(16))))
PY
  done
done
cp /tmp/fe_common.h.keep feinsum_amd/csrc/fe_common.h
