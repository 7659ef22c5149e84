#!/bin/bash
# round 4, call O: the solver leaves out the tiles under snow on mixed days (mcf_plan_run_days_masked): tests, then configs[4]'s share A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04o; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun_gpu.py tests/test_snow_gpu.py tests/test_bench_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -40 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
for v in skip ring4 skip2 ring42; do
  unset MCF_LIB; [ "${v#ring4}" != "$v" ] && export MCF_LIB="$GRAFT_REPO_ROOT/build/variants/libmcfhip_ring4.so"
  MCF_BENCH_STAGES=1 timeout -k 10 600 python bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/$v.json 2> $o/$v.err || { tail -5 $o/$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04o/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.4e" % d["value"], {k: round(v / 2, 3) for k, v in d["stage_seconds"].items()}, d["config"]["solver_tile_days_left_out"][:6])
PY
done
unset MCF_LIB
timeout -k 10 900 python bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_c4.json 2> $o/bench_c4.err || { tail -5 $o/bench_c4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04o/bench_c4.json").read().strip().splitlines()[-1])
print("final", "%.4e" % d["value"], d["ms_per_step"], d["verified"]["ok"], d["verified"]["max_scaled_err"], d["verified"]["cell_steps_by_class"])
PY
