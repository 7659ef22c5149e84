set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zd; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_snow_micro_pipeline_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -2 $out/tests.log
MCF_BENCH_STAGES=1 timeout -k 10 600 python3 bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $out/config4_stages.json 2> $out/config4_stages.err || { tail -30 $out/config4_stages.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/config4_stages.json')); print('value %.4e ms %.0f' % (d['value'], d['ms_per_step'])); print(d['stage_seconds']); print(d['config']['passes']); print(d['verified'])"
timeout -k 10 600 python3 bench.py --config 4 --steps 2 --warmup 1 > $out/config4_share.json 2> $out/config4_share.err || { tail -30 $out/config4_share.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/config4_share.json')); print('value %.4e ms %.0f' % (d['value'], d['ms_per_step']), d.get('cpu_baseline',{}).get('value'))"
