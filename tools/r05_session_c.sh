#!/bin/bash
# Round 5, GPU session C: phase priorities A/B, the evaluate() loop cost, the changed tests
out=gpurun_out/r05c; mkdir -p $out
timeout -k 10 900 python3 tools/phase_ab.py > $out/phase_ab.txt 2>&1; cat $out/phase_ab.txt
timeout -k 10 300 python3 tools/evaluate_loop_cost.py > $out/evaluate_loop_cost.txt 2>&1; cat $out/evaluate_loop_cost.txt
timeout -k 10 900 python -m pytest tests/test_gpu_streams.py tests/test_placement.py tests/test_gpu_bench_line.py -m gpu -x -q > $out/pytest_changed.log 2>&1; tail -5 $out/pytest_changed.log
