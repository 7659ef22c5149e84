#!/bin/bash
# round 4, call T: the final tree — whole GPU suite, smoke(), the driver's default bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04t; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -40 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04t/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["valu"]["frac_of_issue_peak"], d["verified"]["ok"], {k: (v["value"], v["verified"]["ok"]) for k, v in d["secondary"].items()})
PY
