#!/bin/bash
# Which hardware counter tells a slow placement (four written arrays in one physical block) from a fast one (two + two
# across a joint)?  tools/split_pmc.py under rocprofv3, one counter group per pass; per configuration the mean over its
# last 30 dispatches.      bash tools/split_pmc.sh            -> gpurun_out/split_pmc/summary.txt
set -e
repo=$PWD; out=/tmp/split_pmc; keep=$PWD/gpurun_out/split_pmc
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
n=0
while read -r group; do
  [ -z "$group" ] && continue
  n=$((n+1))
  echo "== pass $n: $group"
  timeout -k 5 200 rocprofv3 --pmc $group --kernel-include-regex facemass --kernel-trace --output-format csv -d "$out/p$n" -o pmc -- python3 "$repo/tools/split_pmc.py" 70 \
      > "$out/p$n.out" 2> "$out/p$n.err" || { tail -5 "$out/p$n.err"; echo "pass $n failed"; continue; }
  grep -v amdgpu.ids "$out/p$n.out" | tr '\n' ';'; echo
done <<'GROUPS'
GRBM_GUI_ACTIVE GRBM_EA_BUSY GRBM_TC_BUSY GRBM_UTCL2_BUSY GRBM_TA_BUSY
TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum
TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TCP_TCR_RDRET_STALL_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum
TD_TC_STALL_sum TD_TD_BUSY_sum TD_SPI_STALL_sum TD_WRITE_ACKT_WAVEFRONT_sum
TCC_BUSY_sum TCC_CYCLE_sum TCC_BUBBLE_sum TCC_IB_STALL_sum
TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_STREAMING_REQ_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
TCC_PROBE_sum TCC_PROBE_ALL_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
GROUPS
python3 - "$out" > "$keep/summary.txt" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
cfgs = ["4 below", "2 + 2", "4 above"]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p*/"), key=lambda s: int(s.rstrip("/").split("p")[-1])):
    tag = d.rstrip("/").split("/")[-1]
    if "no joint" in open(out + "/" + tag + ".out").read():
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        by = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "facemass" not in r["Kernel_Name"]:
                continue
            i = int(r["Dispatch_Id"])
            by[i][r["Counter_Name"]] = float(r["Counter_Value"])
            by[i]["dur_us(" + tag + ")"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
        disp = [by[k] for k in sorted(by)][-120:]
        for ci, c in enumerate(cfgs):
            chunk = disp[ci * 40 + 10:(ci + 1) * 40]
            for name in chunk[0]:
                res[name][c] = sum(x[name] for x in chunk) / len(chunk)
print("counter".ljust(48) + "".join(c.rjust(16) for c in cfgs) + "   (2+2)/(4 below)")
for name in sorted(res):
    v = res[name]
    ratio = v["2 + 2"] / v["4 below"] if v["4 below"] else float("nan")
    print(name.ljust(48) + "".join(f"{v[c]:16.5g}" for c in cfgs) + f"   {ratio:8.3f}")
PY
cat "$keep/summary.txt"
