set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
CONFIG=2 tools/ab_bench2.sh gpurun_out/r03b/ab2 r02=build/variants/libmcfhip_r02.so new=- storeonly=build/variants/libmcfhip_storeonly.so nostore=build/variants/libmcfhip_nostore.so r02b=build/variants/libmcfhip_r02.so new2=-
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03b/tests.log 2>&1 || { tail -40 gpurun_out/r03b/tests.log; exit 1; }
tail -3 gpurun_out/r03b/tests.log
