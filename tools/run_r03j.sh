set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03j; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_math_gpu.py tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_layers_gpu.py tests/test_edge_cases_gpu.py tests/test_coarse_forcing_gpu.py tests/test_random_configs_gpu.py tests/test_random_more_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
CONFIG=1 tools/ab_bench2.sh $out/ab1 r02=build/variants/libmcfhip_r02.so new=- new2=-
CONFIG=2 tools/ab_bench2.sh $out/ab2 r02=build/variants/libmcfhip_r02.so new=- new2=-
