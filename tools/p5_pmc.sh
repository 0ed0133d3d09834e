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
# an un-profiled timing of the same launches beside the counter passes (profiled passes hold a lower clock)
for fam in ${FAMS:-grad div facemass}; do
  for np in ${NPS:-56 35}; do
    python3 "$repo/tools/run_family.py" $fam $np 1000000 ${LAUNCHES:-10} > "$out/${fam}_${np}_plain.out" 2>/dev/null || true
  done
done
python3 - "$out" "$keep" "$repo" > "$keep/summary.txt" <<'PY'
import csv, glob, json, os, re, sys, collections
out, keep, repo = sys.argv[1:4]
sys.path.insert(0, repo)
import bench
FLOPS = {("grad", 56): 19824.0, ("div", 56): 19824.0, ("grad", 35): 7980.0, ("div", 35): 7980.0, ("facemass", 56): 4 * (4 * 21 + 2 * 56 * 4 * 21), ("facemass", 35): 17040.0}
for fam in os.environ.get("FAMS", "grad div facemass").split():
    for np_ in [int(x) for x in os.environ.get("NPS", "56 35").split()]:
        vals = collections.defaultdict(list); dur = []; kname = "?"
        for d in sorted(glob.glob(f"{out}/{fam}_{np_}_*/")):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                seen = set()
                for r in csv.DictReader(open(f)):
                    if "fe::" not in r["Kernel_Name"] or "split_probe" in r["Kernel_Name"] or "generic" in r["Kernel_Name"]:
                        continue
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"]); dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
                    kname = r["Kernel_Name"]
        if not dur:
            continue
        m = {k: sum(v) / len(v) for k, v in vals.items()}
        us = sum(dur) / len(dur)
        act = m.get("GRBM_GUI_ACTIVE", 0) / 8
        mhz = act / us if us else 0.0
        plain_ms = None
        try:
            plain_ms = float(re.search(r": ([0-9.]+) ms", open(f"{out}/{fam}_{np_}_plain.out").read()).group(1))
        except Exception:
            pass
        flops = FLOPS.get((fam, np_), 0.0) * 1e6
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        rec = {"family": fam, "Np": np_, "E": 1000000, "kernel": kname, "source_sha": bench.kernel_source_sha(), "launches": len(dur),
               "duration_us_under_pmc": round(us, 2), "duration_us_unprofiled": None if plain_ms is None else round(plain_ms * 1e3, 2),
               "sclk_mhz_under_pmc": round(mhz), "sclk_note": "GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (reads a few % high below 0.3 ms: MI355X_MICROARCH.md, DVFS give-back)",
               "mfma_busy_fraction": round(busy / (act * 1024), 4) if act else None,
               "mfma_busy_cycles_per_simd": round(busy / 1024),
               "tflops_under_pmc": round(flops / us * 1e-6, 2),
               "tflops_unprofiled": None if plain_ms is None else round(flops / plain_ms * 1e-9, 2),
               "fp64_peak_tflops_at_this_clock": round(78.6 * mhz / 2400.0, 2),
               "lds_bank_conflict_fraction": round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4) if m.get("SQ_LDS_IDX_ACTIVE") else None,
               "lds_busy_fraction": round(m["SQ_LDS_IDX_ACTIVE"] / 256.0 / act, 4) if (act and m.get("SQ_LDS_IDX_ACTIVE")) else None,
               "wait_inst_fraction_of_wave_cycles": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else None,
               "wait_any_fraction_of_wave_cycles": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") else None,
               "counters": {k: round(v, 1) for k, v in sorted(m.items())},
               "profile": "rocprofv3 --pmc <two counter groups, one pass each> --kernel-trace -- python3 tools/run_family.py %s %d 1000000 %s (tools/p5_pmc.sh)" % (fam, np_, os.environ.get("LAUNCHES", "10"))}
        json.dump(rec, open(f"{keep}/traffic_{fam}_p{5 if np_ == 56 else 4}.json", "w"), indent=1, sort_keys=True)
        print(f"== {fam} Np={np_}: {kname[:70]}")
        print(f"   duration {us:.1f} us under PMC ({rec['duration_us_unprofiled']} us un-profiled); GRBM active/XCD {act:.0f} cycles -> {mhz:.0f} MHz")
        print(f"   MFMA busy / (active x 1024 SIMDs) = {rec['mfma_busy_fraction']};  {rec['tflops_under_pmc']} TFLOP/s under PMC, {rec['tflops_unprofiled']} un-profiled; "
              f"fp64 peak at this clock {rec['fp64_peak_tflops_at_this_clock']} TFLOP/s")
        print(f"   LDS: busy {rec['lds_busy_fraction']}, bank conflicts {rec['lds_bank_conflict_fraction']} of the LDS cycles; waves: issue-stalled {rec['wait_inst_fraction_of_wave_cycles']}, parked {rec['wait_any_fraction_of_wave_cycles']}")
        for k in sorted(m):
            print(f"   {k:34s} {m[k]:.4g}")
PY
cat "$keep/summary.txt"
