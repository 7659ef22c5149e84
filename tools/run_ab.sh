cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_last
bash tools/ab_bench.sh $out prev1=build/variants/libmcfhip_prev.so new1=- prev2=build/variants/libmcfhip_prev.so new2=-
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py -x -q -m gpu 2>&1 | tail -1
