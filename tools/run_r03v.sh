cd $GRAFT_REPO_ROOT
out=gpurun_out/r03v; mkdir -p $out
V=build/variants
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_layers_gpu.py tests/test_edge_cases_gpu.py tests/test_random_configs_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -2
CONFIG=1 tools/ab_bench2.sh $out/ab1 head=$V/libmcfhip_head.so new=- head2=$V/libmcfhip_head.so new2=-
CONFIG=2 tools/ab_bench2.sh $out/ab2 head=$V/libmcfhip_head.so new=-
CONFIG=1 EXTRA="--array-forcing --ring-days 5" STEPS=5 tools/ab_bench2.sh $out/af head=$V/libmcfhip_head.so new=-
