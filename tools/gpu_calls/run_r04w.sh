#!/bin/bash
# round 4, call W: the N = 2 path over gloo with both ranks on this one GPU (REHEARSAL lines), configs[2] and configs[4] geometry at size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04w; mkdir -p $o
MCF_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --config 2 --rows 1024 --cols 1024 --tsteps 1920 --steps 1 --warmup 0 --no-secondary > $o/rehearsal_config2.json 2> $o/c2.err || { tail -5 $o/c2.err; exit 1; }
MCF_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --config 4 --share 2 --rows 1024 --cols 1024 --tsteps 1920 --steps 1 --warmup 0 > $o/rehearsal_config4.json 2> $o/c4.err || { tail -5 $o/c4.err; exit 1; }
python - <<'PY'
import json
for c in (2, 4):
    d = json.loads(open(f"gpurun_out/r04w/rehearsal_config{c}.json").read().strip().splitlines()[-1])
    print(c, d["n_gpus"], "%.3e" % d["value"], d["verified"]["ok"], d["config"].get("partition", d["config"].get("halo"))[:90])
PY
