#!/bin/bash
# round 4, call Y: after the terrain stencils — whole GPU suite, configs[4]'s share (stage seconds, then the full line), its kernels under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04y; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -40 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
MCF_BENCH_STAGES=1 timeout -k 10 600 python bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/stages.json 2> $o/stages.err || { tail -5 $o/stages.err; exit 1; }
timeout -k 10 900 python bench.py --config 4 --share 8 --steps 3 --warmup 1 > $o/bench_c4.json 2> $o/bench_c4.err || { tail -5 $o/bench_c4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04y/stages.json").read().strip().splitlines()[-1])
print("stages", "%.4e" % d["value"], {k: round(v / 2, 3) for k, v in d["stage_seconds"].items()})
d = json.loads(open("gpurun_out/r04y/bench_c4.json").read().strip().splitlines()[-1])
print("line", "%.4e" % d["value"], d["ms_per_step"], d["verified"]["ok"], d["verified"]["max_scaled_err"])
PY
rm -rf gpurun_out/prof_r04_c4_aux
bash tools/profile_aux.sh r04_c4 bench.py --config 4 --share 8 --steps 1 --warmup 1 --no-cpu-baseline --no-verify
