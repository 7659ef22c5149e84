cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zh; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in shipped:- w43:build/variants/libmcfhip_snow_w43.so; do
  n=${v%%:*}; l=${v#*:}
  ( [ "$l" != "-" ] && export MCF_LIB=$PWD/$l; rocprofv3 --kernel-trace --stats --output-format csv -d $out/$n -- python3 tools/aux_kernels_workload.py > $out/$n.log 2> $out/$n.err ) || echo "$n failed"
  grep -h "k_snowmodel\|k_microsnow<" $out/$n/*/*kernel_stats.csv | cut -d, -f1-4
done
