#!/bin/bash
# round 4, call R: configs[4]'s share on the final tree (with the in-run verification and the CPU baseline)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04r; mkdir -p $o
timeout -k 10 1000 python bench.py --config 4 --share 8 --steps 3 --warmup 1 > $o/bench_c4.json 2> $o/bench_c4.err || { tail -20 $o/bench_c4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04r/bench_c4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"]["ok"], d["verified"]["max_scaled_err"], d["verified"]["snowmodel"]["ok"], d["cpu_baseline"]["value"])
print(d["config"]["solver_tile_days_left_out"][:40], "|", d["config"]["passes"][-330:])
PY
