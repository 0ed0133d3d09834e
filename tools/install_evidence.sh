#!/bin/bash
# Copies what one evidence cycle left under gpurun_out/ into profiles/ (run here, after the GPU call has merged its outputs):
#   gpurun -- 'bash tools/profile_round.sh r05 && ELEMS=100000 bash tools/profile_round.sh r05 grad div graddiv pipeline &&
#              cp gpurun_out/prof_r05/traffic_*.json profiles/ && bash tools/final_r05.sh; bash tools/final_r05_b.sh'
#   bash tools/install_evidence.sh
set -e
cd "$(dirname "$0")/.."
e=gpurun_out
cp $e/prof_r05/traffic_*.json profiles/
cp $e/prof_r05/*_kernel_stats.csv $e/prof_r05/bench_*_under_rocprof.json profiles/r05/
cp $e/final_r05/bench_*.json profiles/r05/
cp $e/final_r05/summary.txt profiles/r05/summary.txt
tail -8 $e/final_r05/pytest_gpu.log > profiles/r05/pytest_gpu_tail.txt
cp $e/final_r05/smoke.log profiles/r05/smoke.txt
o=$e/final_r05_b
cp $o/bench_*_box2.json profiles/r05/
{ cat $o/fuzz_gpu.txt; sed -n '/# three more sweeps/,$p' profiles/r05/fuzz_gpu_final.txt; } > /tmp/_fuzz.txt && cp /tmp/_fuzz.txt profiles/r05/fuzz_gpu_final.txt
grep -E "^==|duration|MFMA busy|LDS:" $o/p5_pmc.log > profiles/r05/p5_pmc_summary.txt
cp $o/traffic_*_p5.json profiles/
cp $o/rehearse.txt profiles/r05/rehearse_ranks.txt
cp $o/selfspawn6.json profiles/r05/rehearse6_selfspawn6.json
cp $o/torchrun4_pipeline_gather.json profiles/r05/rehearse4_torchrun4_pipeline_gather.json
for fam in grad div; do
  cp $o/stamps_${fam}_100000.csv.tiles.csv profiles/r05/stamps_${fam}_100000_tiles_final.csv
  cp $o/tiles_${fam}_100000.txt profiles/r05/tiles_${fam}_100000_final.txt
done
tail -2 profiles/r05/pytest_gpu_tail.txt
python3 tools/roofline_table.py profiles/r05
