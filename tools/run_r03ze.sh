set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03ze; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_dispatch_gpu.py tests/test_parity_gpu.py tests/test_coarse_forcing_gpu.py tests/test_random_configs_gpu.py tests/test_edge_cases_gpu.py tests/test_multi_device_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for i in 1 2; do
python3 bench.py --config 1 --array-forcing --ring-days 5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $out/af_$i.json 2>> $out/err.txt
python3 bench.py --config 1 --coarse 8x8 --ring-days 5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $out/coarse_$i.json 2>> $out/err.txt
done
python3 bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-verify > $out/c1.json 2>> $out/err.txt
python3 -c "
import json
for f in ['af_1','af_2','coarse_1','coarse_2','c1']:
    d=json.load(open('$out/'+f+'.json')); print(f, '%.4e'%d['value'], d['roofline']['avg_launch_ms'])"
