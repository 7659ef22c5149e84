set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03u; mkdir -p $out
V=build/variants
{
echo "[r03 config 1] 1024^2 x 8760 h, 10-day launches, same box: round-2 kernel / shipped / store-only / stores elided / stores into an L2-resident window / no barrier / prologue only / persistent loop"
CONFIG=1 tools/ab_bench2.sh $out/ab1 r02=$V/libmcfhip_r02.so shipped=- storeonly=$V/libmcfhip_storeonly.so nostore=$V/libmcfhip_nostore.so storehot=$V/libmcfhip_storehot.so nobarrier=$V/libmcfhip_nobarrier.so prologue=$V/libmcfhip_prologue_only.so persist=$V/libmcfhip_persistent_loop.so sections=$V/libmcfhip_sections.so shipped2=-
grep "mcf sections" $out/ab1/sections.err | tail -9
echo "[r03 config 2] 4096^2 x 8760 h, device terrain, 7-day launches, same box"
CONFIG=2 tools/ab_bench2.sh $out/ab2 r02=$V/libmcfhip_r02.so shipped=- storeonly=$V/libmcfhip_storeonly.so nostore=$V/libmcfhip_nostore.so prologue=$V/libmcfhip_prologue_only.so shipped2=-
} | tee $out/timing_experiments.txt
