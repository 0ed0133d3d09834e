#!/bin/bash
# What differs between a slow and a fast placement of the same launch?  The placement sweep (explicit gaps,
# 120 dispatches each) under rocprofv3, one counter group per pass; per gap the mean of every counter.
#   bash tools/placement_pmc.sh <facemass|grad> "<gaps MiB, comma separated>"
set -e
fam=${1:-facemass}; gaps=${2:-0,296,616,776,1040,1248}
repo=$PWD; out=/tmp/placement_pmc; keep=$PWD/gpurun_out/placement_pmc_$fam
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
export FE_GAPS=$gaps
n=0
for group in \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
  "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  n=$((n+1))
  echo "== pass $n: $group"
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/p$n" -o pmc -- python3 "$repo/tools/placement_sweep.py" $fam gap 1000000 wide \
      > "$out/p$n.out" 2> "$out/p$n.err" || { tail -5 "$out/p$n.err"; exit 1; }
  grep "median" "$out/p$n.out" > "$keep/times_pass$n.txt" || true
done
python3 - "$out" "$gaps" > "$keep/summary.txt" <<'PY'
import csv, sys, glob, collections
out, gaps = sys.argv[1], sys.argv[2].split(",")
per = 120
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "mfma_kernel" in r["Kernel_Name"]]
        by_disp = collections.defaultdict(dict)
        for r in rows:
            by_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            by_disp[int(r["Dispatch_Id"])]["_dur_us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
        disp = [by_disp[k] for k in sorted(by_disp)]
        for gi, gap in enumerate(gaps):
            chunk = disp[gi * per + 20:(gi + 1) * per]          # skip the warm-up launches
            if not chunk:
                continue
            for c in chunk[0]:
                res[gap][c if c != "_dur_us" else "dur_us(" + d.rstrip("/").split("/")[-1] + ")"] = sum(x[c] for x in chunk) / len(chunk)
names = sorted({c for g in res.values() for c in g})
print("counter".ljust(44) + "".join(f"gap {g:>6}".rjust(16) for g in gaps))
for c in names:
    print(c.ljust(44) + "".join(f"{res[g].get(c, float('nan')):16.4g}" for g in gaps))
PY
cat "$keep/summary.txt"
