#!/bin/bash
# round 4, call N: the snow kernels' step tables through scalar loads (restrict kernel arguments): tests, then stage seconds of configs[4]'s
# share with the ring kernel built for 3 (tree) and 4 waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04n; mkdir -p $o
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_snow_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun_gpu.py tests/test_random_snow_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -30 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
fi
for v in tree ring4 tree2 ring42; do
  unset MCF_LIB; [ "${v#ring4}" != "$v" ] && export MCF_LIB="$GRAFT_REPO_ROOT/build/variants/libmcfhip_ring4.so"
  MCF_BENCH_STAGES=1 timeout -k 10 600 python bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/$v.json 2> $o/$v.err || { tail -5 $o/$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04n/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.4e" % d["value"], {k: round(v / 2, 3) for k, v in d["stage_seconds"].items()})
PY
done
