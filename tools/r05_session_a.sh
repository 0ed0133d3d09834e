#!/bin/bash
# Round 5, GPU session A: measurements on the sources of the round's start (+ the wide-block experiment):
#   the MFMA / VALU overlap microbenchmark with its bf16 control, the sixteen-waves-per-CU A/B, PMC passes at E = 1e5, p = 5 counters
out=gpurun_out/r05a; mkdir -p $out
{
for b in mfma_valu_overlap mfma_valu_overlap_f32 mfma_valu_overlap_bf16_32 mfma_valu_overlap_bf16_16; do timeout -k 10 120 build/$b; echo; done
} > $out/mfma_valu_overlap_with_control.txt 2>&1
tail -5 $out/mfma_valu_overlap_with_control.txt
timeout -k 10 600 python3 tools/wide_ab.py > $out/wide_ab.txt 2>&1; tail -20 $out/wide_ab.txt
ELEMS=100000 timeout -k 10 900 bash tools/profile_round.sh r05a grad div > $out/profile_1e5.log 2>&1; tail -4 $out/profile_1e5.log
FAMS="grad div" NPS=56 LAUNCHES=300 timeout -k 10 600 bash tools/p5_pmc.sh > $out/p5_pmc.log 2>&1; tail -40 $out/p5_pmc.log
