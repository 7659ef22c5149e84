cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_lines; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python3 - <<P
import json
d=json.load(open("$out/bench_default.json"))
print("value %.4e  ms/step %.1f  frac %.3f  verified %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verified"] and (d["verified"]["ok"], d["verified"]["max_scaled_err"])))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], "traffic", d["roofline"]["traffic"], d["roofline"].get("counters"))
for k,v in d["secondary"].items(): print(k, v.get("value"), v.get("hbm_frac"), v.get("avg_launch_ms"), v.get("error"))
P
python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/config3_share.json 2> $out/config3_share.err || tail -5 $out/config3_share.err
python3 bench.py --config 4 --steps 2 --warmup 1 > $out/config4_share.json 2> $out/config4_share.err || tail -5 $out/config4_share.err
MCF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --config 2 --rows 1024 --cols 1024 --tsteps 1920 --steps 1 --warmup 0 --no-secondary > $out/rehearsal_config2.json 2> $out/rehearsal_config2.err || tail -5 $out/rehearsal_config2.err
MCF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --config 4 --rows 1024 --cols 1024 --share 2 --tsteps 1920 --steps 1 --warmup 0 > $out/rehearsal_config4.json 2> $out/rehearsal_config4.err || tail -5 $out/rehearsal_config4.err
for f in config3_share config4_share rehearsal_config2 rehearsal_config4; do python3 -c "
import json; d=json.load(open('$out/$f.json')); print('$f', '%.4e'%d['value'], d['n_gpus'], 'cpu' in str(d.get('cpu_baseline'))[:4] or d.get('cpu_baseline',{}).get('value'), (d.get('verified') or {}).get('ok'), d['config'].get('halo','')[:60], d['config'].get('partition','')[-90:])"; done
