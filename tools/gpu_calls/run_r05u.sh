#!/bin/bash
# round 5, call U: what bounds the snow-day microclimate kernel — timing variants (results wrong on purpose): no value stored / the
# loads and stores without the physics; stage times of configs[4]'s share, kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && o=gpurun_out/r05u && mkdir -p $o
for m in shipped nostore nophysics; do
  case $m in shipped) unset MCF_LIB;; *) export MCF_LIB=$GRAFT_REPO_ROOT/build/variants/libmcfhip_ms_$m.so;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/$m -- python3 bench.py --config 4 --share 8 --steps 1 --warmup 1 --no-verify --no-cpu-baseline > $o/$m.json 2> $o/$m.err || exit 1
  python3 - <<P
import csv, glob
for f in glob.glob("$o/$m/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_microsnow", "k_solve<", "k_snowmodel")):
            print("%-10s %-50s calls %s avg %.3f ms" % ("$m", r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
P
done
