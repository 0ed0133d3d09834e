#!/bin/bash
# Round 5, GPU session G: the interleaved div in the fused launches; tickets from three rounds in fused launches; more sizes of plain div
out=gpurun_out/r05g; mkdir -p $out
timeout -k 10 600 python3 tools/phase_ab.py graddiv pipeline knob=ilv 20000 50000 80000 98304 100000 131072 160000 200000 400000 1000000 > $out/fused_interleave_ab.txt 2>&1; cat $out/fused_interleave_ab.txt
timeout -k 10 600 python3 tools/phase_ab.py graddiv pipeline knob=tickets3 ilv=on 80000 98304 100000 110000 120000 131072 140000 160000 > $out/fused_tickets3_ab.txt 2>&1; cat $out/fused_tickets3_ab.txt
timeout -k 10 600 python3 tools/phase_ab.py div knob=ilv 10000 30000 40000 50000 60000 70000 > $out/div_interleave_small.txt 2>&1; cat $out/div_interleave_small.txt
timeout -k 10 600 python3 tools/phase_ab.py grad div knob=tickets3 ilv=on 98304 100000 110000 120000 131072 > $out/single_tickets3_ab.txt 2>&1; cat $out/single_tickets3_ab.txt
