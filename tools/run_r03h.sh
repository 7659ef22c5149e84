set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03h; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
CONFIG=1 tools/ab_bench2.sh $out/ab1 warm0=-,MCF_WARM_AHEAD=0 warm64=- warm32=-,MCF_WARM_AHEAD=32 warm128=-,MCF_WARM_AHEAD=128 warm0b=-,MCF_WARM_AHEAD=0 sect0=build/variants/libmcfhip_sections.so,MCF_WARM_AHEAD=0 sect64=build/variants/libmcfhip_sections.so
grep "mcf sections" $out/ab1/sect0.err | tail -9
grep "mcf sections" $out/ab1/sect64.err | tail -9
CONFIG=2 tools/ab_bench2.sh $out/ab2 warm0=-,MCF_WARM_AHEAD=0 warm64=- warm0b=-,MCF_WARM_AHEAD=0 warm64b=-
