#!/bin/bash
# Round 5, GPU session D: arbiter microbenchmark, div with the interleaved B build, p = 5 phase priorities, changed tests, p = 5 counters
out=gpurun_out/r05d; mkdir -p $out
{ for b in mfma_valu_overlap mfma_valu_overlap_bf16_32; do timeout -k 10 120 build/$b | grep -A14 "pure MFMA\|^# v_mfma"; echo; done; } > $out/mfma_arbiter.txt 2>&1; cat $out/mfma_arbiter.txt
timeout -k 10 600 python3 tools/phase_ab.py div knob=ilv > $out/div_interleave_ab.txt 2>&1; cat $out/div_interleave_ab.txt
timeout -k 10 600 python3 tools/p5_phase_ab.py > $out/p5_phase_ab.txt 2>&1; cat $out/p5_phase_ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_streams.py tests/test_placement.py tests/test_gpu_bench_line.py -m gpu -x -q > $out/pytest_changed.log 2>&1; tail -5 $out/pytest_changed.log
FAMS="grad div" NPS=56 LAUNCHES=300 timeout -k 10 600 bash tools/p5_pmc.sh > $out/p5_pmc.log 2>&1; tail -12 $out/p5_pmc.log; cp gpurun_out/p5_pmc/traffic_*_p5.json $out/ 2>/dev/null
