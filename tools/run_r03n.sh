set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03n; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_edge_cases_gpu.py tests/test_random_configs_gpu.py tests/test_coarse_forcing_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
CONFIG=1 EXTRA="--array-forcing --ring-days 5" STEPS=5 tools/ab_bench2.sh $out/af r02=build/variants/libmcfhip_r02.so new32=- 
CONFIG=1 EXTRA="--coarse 8x8 --ring-days 5" STEPS=3 tools/ab_bench2.sh $out/af r02coarse=build/variants/libmcfhip_r02.so coarse32=-
