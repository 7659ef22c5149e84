set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03p; mkdir -p $out
MCF_BENCH_STAGES=1 timeout -k 10 900 python3 bench.py --config 4 --steps 1 --warmup 0 > $out/config4_stages.json 2> $out/config4_stages.err || { tail -30 $out/config4_stages.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/config4_stages.json')); print('value %.4e ms %.0f' % (d['value'], d['ms_per_step'])); print(d['stage_seconds']); print(d['config']['solver_days_per_year'], d['config']['snow_days_per_year'])"
timeout -k 10 900 python3 bench.py --config 4 --steps 2 --warmup 1 > $out/config4.json 2> $out/config4.err || { tail -30 $out/config4.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/config4.json')); print('value %.4e ms %.0f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
