#!/bin/bash
# All round-end measurements in one GPU call; outputs under gpurun_out/summary_<tag>/ (copy to profiles/<round>/).
tag=${1:-r01}
out=gpurun_out/summary_$tag
mkdir -p $out
python bench.py > $out/bench_grad.json 2> $out/bench_grad.err
for w in div facemass graddiv pipeline; do python bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2>> $out/bench_grad.err; done
python tools/bench_batched.py > $out/bench_batched.txt 2>&1
python tools/bench_siblings.py > $out/bench_siblings.txt 2>&1
python tools/bench_archive_families.py > $out/bench_archive_families.txt 2>&1
python tools/bench_tiled.py > $out/bench_tiled.txt 2>&1
for f in grad div facemass graddiv; do for np in 20 10 4; do echo -n "Np=$np "; FE_NP=$np ./build/fe_check $f 1000000 0 30 1 | tail -1; done; done > $out/lower_orders_fe_check.txt 2>&1
for E in 20000 100000 100003 1000000 1000007 8000000; do ./build/fe_check grad $E 0 50 1 | tail -1; done > $out/grad_sizes_fe_check.txt 2>&1
tail -n +1 $out/*.txt | grep -v amdgpu.ids
cat $out/bench_*.json | cut -c1-400
