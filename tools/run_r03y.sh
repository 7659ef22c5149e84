set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03y; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_snow_gpu.py tests/test_random_snow_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowfast_gpu.py tests/test_frontend_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -2 $out/tests.log
MCF_BENCH_STAGES=1 timeout -k 10 600 python3 bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $out/config4_stages.json 2> $out/config4_stages.err || { tail -30 $out/config4_stages.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/config4_stages.json')); print('value %.4e ms %.0f' % (d['value'], d['ms_per_step'])); print(d['stage_seconds'])"
timeout -k 10 900 bash tools/profile_aux.sh r03y > $out/aux.log 2>&1 || { tail -20 $out/aux.log; exit 1; }
python3 tools/summarize_aux.py r03y > $out/aux_summary.txt 2>&1 || tail -5 $out/aux_summary.txt
grep -i "snowmodel\|microsnow" $out/aux_summary.txt | head
