#!/bin/bash
# A/B of the snow kernels on the GPU box: the aux workload under rocprofv3 --stats, once per library variant.
# usage: tools/ab_snow.sh <outdir> name=lib ...      (lib "-" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  [ "$lib" != "-" ] && export MCF_LIB=$GRAFT_REPO_ROOT/$lib || unset MCF_LIB
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 tools/aux_kernels_workload.py > $out/$name.log 2> $out/$name.err
  python3 - <<P
import csv, glob
for f in glob.glob("$out/$name/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_snowmodel", "k_microsnow")):
            print("%-12s %-60s calls %s avg %.3f ms" % ("$name", r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
P
done
