set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03g; mkdir -p $out
CONFIG=1 tools/ab_bench2.sh $out/ab1 new=- sections=build/variants/libmcfhip_sections.so
grep "mcf sections" $out/ab1/sections.err | tail -9
CONFIG=2 STEPS=1 tools/ab_bench2.sh $out/ab2 sections=build/variants/libmcfhip_sections.so
grep "mcf sections" $out/ab2/sections.err | tail -9
