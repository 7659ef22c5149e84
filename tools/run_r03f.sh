set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03f; mkdir -p $out
CONFIG=1 tools/ab_bench2.sh $out/ab1 new=- persist=build/variants/libmcfhip_persist.so persist1024=build/variants/libmcfhip_persist.so,MCF_PERSIST_WGS=1024 new2=-
CONFIG=2 tools/ab_bench2.sh $out/ab2 new=- persist=build/variants/libmcfhip_persist.so
