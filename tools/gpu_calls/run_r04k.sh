#!/bin/bash
# round 4, call K: the pitched-download change (tests, then the sink rates of call I again), the driver's default bench and the configs[4] share
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04k; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_multi_device_gpu.py tests/test_snowrun_gpu.py "tests/test_snow_gpu.py" -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -30 $o/pytest.txt; exit 1; }
tail -3 $o/pytest.txt
{
echo "== mcf_runmicro1_multi (Tz only into a numpy array) and mcf_runbioclim1_multi, 4096 x 4096, 4 row blocks time-sliced on ONE device"
timeout -k 10 900 python tools/multi_rate.py --rows 4096 --cols 4096 --tsteps 240 --devices 0 --blocks 4 --what solver,bioclim
echo "== the same with one block (the single-device call through the same driver)"
timeout -k 10 900 python tools/multi_rate.py --rows 4096 --cols 4096 --tsteps 240 --devices 0 --blocks 1 --what solver,bioclim
echo "== hipMemcpy2D instead of the pinned ring (MCF_NO_HOSTPIPE=1), 4 blocks"
MCF_NO_HOSTPIPE=1 timeout -k 10 900 python tools/multi_rate.py --rows 4096 --cols 4096 --tsteps 240 --devices 0 --blocks 4 --what solver
echo "== mcf_snowmodel1_multi: host-side price of the scatter"
timeout -k 10 900 python tools/snow_multi_cmp.py
} > $o/sink_rates.txt 2>&1
grep -v amdgpu.ids $o/sink_rates.txt
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err && tail -c 1200 $o/bench_default.json
