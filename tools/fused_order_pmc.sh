#!/bin/bash
# HBM request mix of the fused div + grad launch with the bodies in the same order on every block (FE_FUSED_ORDER=0) and
# with the younger half of the grid running grad first (1): bench.py through the experiment build under rocprofv3.
set -e
repo=$PWD; out=/tmp/fo_pmc; keep=$PWD/gpurun_out/fused_order_pmc
rm -rf "$out" && mkdir -p "$out" "$keep"
export FEINSUM_HIP_LIB=$repo/build/libfeinsum_hip_exp.so
cd /tmp && export TMPDIR=/tmp
n=0
for group in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_WRITEBACK_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  for o in 0 1; do
    export FE_FUSED_ORDER=$o
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/p${n}_o$o" -o pmc -- python3 "$repo/bench.py" --workload graddiv \
        --no-cpu-baseline --no-protocol --placement separate --setup-launches 3 --steps 10 --warmup 2 > /dev/null 2> "$out/err.txt" || { tail -3 "$out/err.txt"; exit 1; }
    echo "pass $n order $o done"
  done
done
python3 - "$out" > "$keep/summary.txt" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p*_o*/")):
    o = d.rstrip("/")[-1]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "graddiv" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            res[c][o] = sum(v) / len(v)
print("counter".ljust(40) + "order 0".rjust(16) + "order 1".rjust(16) + "   ratio")
for c in sorted(res):
    a, b = res[c].get("0", float("nan")), res[c].get("1", float("nan"))
    print(c.ljust(40) + f"{a:16.6g}{b:16.6g}   {b / a if a else float('nan'):.4f}")
PY
cat "$keep/summary.txt"
