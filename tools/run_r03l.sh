set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03l; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_edge_cases_gpu.py tests/test_random_configs_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
bash tools/pmc_valu.sh $out/c2
bash tools/pmc_valu.sh $out/c1 --config 1
CONFIG=2 tools/ab_bench2.sh $out/ab2 r02=build/variants/libmcfhip_r02.so new=-
