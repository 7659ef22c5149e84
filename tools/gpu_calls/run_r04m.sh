#!/bin/bash
# round 4, call M: stage seconds of the configs[4] share on the final tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04m; mkdir -p $o
MCF_BENCH_STAGES=1 timeout -k 10 1000 python bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/bench_c4_stages.json 2> $o/bench_c4.err || { tail -20 $o/bench_c4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04m/bench_c4_stages.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("stage_seconds"), indent=0))
PY
