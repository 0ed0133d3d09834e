#!/bin/bash
# MFMA / LDS counters of the p = 5 kernels (and p = 4 for comparison):  bash tools/p5_pmc.sh
set -e
repo=$PWD; out=/tmp/p5_pmc; keep=$PWD/gpurun_out/p5_pmc
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"
for fam in ${FAMS:-grad div facemass}; do
  for np in ${NPS:-56 35}; do
    n=0
    for group in "$G1" "$G2"; do
      n=$((n+1))
      rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/${fam}_${np}_$n" -o pmc -- python3 "$repo/tools/run_family.py" $fam $np 1000000 ${LAUNCHES:-10} \
        > "$out/${fam}_${np}_$n.out" 2> "$out/${fam}_${np}_$n.err" || { tail -5 "$out/${fam}_${np}_$n.err"; exit 1; }
    done
  done
done
python3 - "$out" > "$keep/summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
import os
for fam in os.environ.get("FAMS", "grad div facemass").split():
    for np_ in [int(x) for x in os.environ.get("NPS", "56 35").split()]:
        vals = collections.defaultdict(list); dur = []
        for d in sorted(glob.glob(f"{out}/{fam}_{np_}_*/")):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                seen = set()
                for r in csv.DictReader(open(f)):
                    if "fe::" not in r["Kernel_Name"] or "split_probe" in r["Kernel_Name"] or "generic" in r["Kernel_Name"]:
                        continue
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"]); dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
                    kname = r["Kernel_Name"][:70]
        m = {k: sum(v) / len(v) for k, v in vals.items()}
        act = m.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f"== {fam} Np={np_}: {kname}")
        print(f"   duration {sum(dur)/len(dur):.1f} us (under PMC); GRBM active/XCD {act:.0f} cycles -> {act/(sum(dur)/len(dur)):.0f} MHz")
        if act:
            print(f"   MFMA busy / (active x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES']/(act*1024):.3f}")
        for k in sorted(m):
            print(f"   {k:34s} {m[k]:.4g}")
PY
cat "$keep/summary.txt"
