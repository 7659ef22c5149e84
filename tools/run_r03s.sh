set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03s; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_coarse_forcing_gpu.py tests/test_dispatch_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
CONFIG=1 EXTRA="--coarse 8x8 --ring-days 5" STEPS=3 tools/ab_bench2.sh $out/af r02coarse=build/variants/libmcfhip_r02.so coarse32=-
bash tools/pmc_valu.sh $out/coarse_pmc --config 1 --coarse 8x8 --ring-days 5
