#!/bin/bash
# round 5, call F: the snow-day microclimate kernels without the device libm's pow / sin / cos on their cold paths (no scratch at
# four waves per SIMD): snow tests, both shapes' kernel times, HBM counters of both shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05f; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_snow_gpu.py tests/test_random_snow_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun_gpu.py tests/test_snowfast_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -5 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
W="bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for v in tiles ring; do
  case $v in ring*) export MCF_MICRORING_OLD=1;; *) unset MCF_MICRORING_OLD;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$v -- python3 $W > $o/$v.json 2> $o/$v.err
  rocprofv3 --kernel-trace --output-format csv -d $o/pmc_fetch_$v --pmc FETCH_SIZE -- python3 $W > /dev/null 2> $o/pmc_fetch_$v.err
  rocprofv3 --kernel-trace --output-format csv -d $o/pmc_write_$v --pmc WRITE_SIZE -- python3 $W > /dev/null 2> $o/pmc_write_$v.err
  python3 - <<P
import csv, glob, json, collections
d = json.load(open("$o/$v.json"))
print("%-6s year %.1f ms (under rocprof)" % ("$v", d["ms_per_step"]))
for f in glob.glob("$o/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_microsnow", "k_snowmodel")):
            print("%-6s %-60s calls %5s avg %8.3f ms total %8.1f ms" % ("$v", r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
for dd in ("pmc_fetch_$v", "pmc_write_$v"):
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % dd, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_microsnow" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][25:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print("%-6s" % "$v", k, "launches", len(v), "mean GB %.2f max GB %.2f" % (sum(v) / len(v) * 1024 / 1e9, max(v) * 1024 / 1e9), "(FETCH_SIZE: x2 on gfx950)")
P
done 2>&1 | tee $o/stats.txt
