set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_edge_cases_gpu.py tests/test_packed_sink_gpu.py tests/test_bioclim_gpu.py tests/test_layers_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_coarse_forcing_gpu.py -x -q -m gpu > gpurun_out/r03a/tests.log 2>&1 || { tail -40 gpurun_out/r03a/tests.log; exit 1; }
tail -3 gpurun_out/r03a/tests.log
CONFIG=1 tools/ab_bench2.sh gpurun_out/r03a/ab1 r02=build/variants/libmcfhip_r02.so new=- storeonly=build/variants/libmcfhip_storeonly.so nostore=build/variants/libmcfhip_nostore.so storehot=build/variants/libmcfhip_storehot.so prologue=build/variants/libmcfhip_prologue_only.so r02b=build/variants/libmcfhip_r02.so new2=-
