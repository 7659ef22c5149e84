# the round's closing run (through gpurun): every GPU test, smoke(), the driver's bench command
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_final; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
if [ "$1" = "bench" ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python3 - <<P
import json
d=json.load(open("$out/bench_default.json"))
print("value %.4e  ms/step %.1f  frac %.3f  verified %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verified"] and (d["verified"]["ok"], d["verified"]["max_scaled_err"])))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], "traffic", d["roofline"]["traffic"], d["roofline"].get("counters"))
for k,v in d["secondary"].items(): print(k, v.get("value"), v.get("hbm_frac"), v.get("avg_launch_ms"), v.get("error"))
P
fi
