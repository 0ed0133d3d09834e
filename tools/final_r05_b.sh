out=gpurun_out/final_r05_b; mkdir -p $out
bash tools/rehearse_ranks.sh $out > $out/rehearse.txt 2>&1; tail -14 $out/rehearse.txt
FAMS="grad div" NPS=56 LAUNCHES=300 timeout -k 10 600 bash tools/p5_pmc.sh > $out/p5_pmc.log 2>&1; grep -E "^==|duration|MFMA busy|LDS:" $out/p5_pmc.log; cp gpurun_out/p5_pmc/traffic_*_p5.json $out/
for fam in grad div; do
  v=1032; [ $fam = div ] && v=1128
  FE_DIV_ILV=1 FE_DUMP_STAMPS=$out/stamps_${fam}_100000.csv timeout -k 10 120 build/fe_check_exp ab $fam 100000 5 50 0,$v > $out/stamps_${fam}_100000.txt 2>&1
  python3 tools/tile_stamps_report.py $out/stamps_${fam}_100000.csv.tiles.csv $fam > $out/tiles_${fam}_100000.txt 2>&1
done
timeout -k 10 400 python3 tools/fuzz_gpu.py > $out/fuzz_gpu.txt 2>&1; tail -3 $out/fuzz_gpu.txt
for w in grad div graddiv pipeline; do python3 bench.py --workload $w --elems-per-gpu 100000 --no-cpu-baseline > $out/bench_${w}_1e5_box2.json 2>> $out/bench.err; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_grad_driver_box2.json 2>> $out/bench.err
for f in $out/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1].split("/")[-1], "value %.0f" % d["value"], "kernel_ms", d.get("kernel_ms"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "mfma_util", d.get("mfma_util"), "at launch time", d.get("mfma_util_at_this_launch_time"))
PY
done
