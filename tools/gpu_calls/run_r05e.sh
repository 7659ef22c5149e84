#!/bin/bash
# round 5, call E: the snow-day microclimate kernels built for THREE waves per SIMD (141 VGPRs, no scratch) against the four-wave
# builds (128 VGPRs + 44 / 48 B of scratch per lane), both shapes; counters of the three-wave tile shape; the layered snow run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snowrun_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -5 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
W="bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for v in tiles4 ring4 tiles3 ring3; do
  case $v in ring*) export MCF_MICRORING_OLD=1;; *) unset MCF_MICRORING_OLD;; esac
  case $v in *3) export MCF_LIB=$PWD/build/variants/libmcfhip_ring3.so;; *) unset MCF_LIB;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$v -- python3 $W > $o/$v.json 2> $o/$v.err
  python3 - <<P
import csv, glob, json
d = json.load(open("$o/$v.json"))
print("%-7s year %.1f ms" % ("$v", d["ms_per_step"]))
for f in glob.glob("$o/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_microsnow",)):
            print("%-7s %-60s calls %5s avg %8.3f ms total %8.1f ms" % ("$v", r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
P
done 2>&1 | tee $o/stats.txt
unset MCF_MICRORING_OLD
export MCF_LIB=$PWD/build/variants/libmcfhip_ring3.so
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
