#!/bin/bash
# round 5, call R: kernel statistics of one configs[4] year with cells / with tiles as the unit of what the both-class days leave out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && o=gpurun_out/r05r && mkdir -p $o
for m in tiles cells; do
  case $m in tiles*) export MCF_SNOW_NO_CELL_GATHER=1;; *) unset MCF_SNOW_NO_CELL_GATHER;; esac
  rocprofv3 --kernel-trace --stats -d $o/$m -o run -- python3 bench.py --config 4 --share 8 --steps 1 --warmup 1 --no-verify --no-cpu-baseline > $o/$m.json 2> $o/$m.err || exit 1
  echo "== $m"; python3 - <<P
import csv, glob
f = glob.glob("$o/$m/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%10.3f ms total" % (float(r["TotalDurationNs"]) / 1e6), "%9.3f avg" % (float(r["AverageNs"]) / 1e6))
P
done
