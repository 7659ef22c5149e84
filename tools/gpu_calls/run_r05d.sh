#!/bin/bash
# round 5, call D: k_microsnow_tiles vs k_microsnow_ring — kernel times (rocprofv3 --stats) and HBM counters of the new shape, on
# configs[4]'s one-rank share
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05d; mkdir -p $o
W="bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for v in new old; do
  case $v in old) export MCF_MICRORING_OLD=1;; *) unset MCF_MICRORING_OLD;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$v -- python3 $W > $o/$v.json 2> $o/$v.err
  python3 - <<P
import csv, glob
for f in glob.glob("$o/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_snowmodel", "k_microsnow", "k_solve<")):
            print("%-4s %-70s calls %5s avg %8.3f ms total %8.1f ms" % ("$v", r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
P
done 2>&1 | tee $o/stats.txt
unset MCF_MICRORING_OLD
rocprofv3 --kernel-trace --output-format csv -d $o/pmc_fetch --pmc FETCH_SIZE -- python3 $W > /dev/null 2> $o/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv -d $o/pmc_write --pmc WRITE_SIZE -- python3 $W > /dev/null 2> $o/pmc_write.err
python3 - <<P | tee $o/pmc.txt
import csv, glob, collections
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_microsnow" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(k, "launches", len(v), "mean KB", sum(v) / len(v), "max KB", max(v))
P
