cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zg
bash tools/ab_bench.sh $out prev1=build/variants/libmcfhip_prev.so new1=- prev2=build/variants/libmcfhip_prev.so new2=-
EXTRA="--array-forcing --ring-days 10 --no-verify" STEPS=3 bash tools/ab_bench.sh $out/af prev=build/variants/libmcfhip_prev.so new=-
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_packed_sink_gpu.py tests/test_random_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
