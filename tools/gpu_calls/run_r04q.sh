#!/bin/bash
# round 4, call Q: the whole GPU suite on the tree, then configs[4]'s share with stage seconds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04q; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1 || { tail -40 $o/pytest.txt; exit 1; }
tail -2 $o/pytest.txt
for v in a b; do
  MCF_BENCH_STAGES=1 timeout -k 10 600 python bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/$v.json 2> $o/$v.err || { tail -5 $o/$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04q/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.4e" % d["value"], {k: round(v / 2, 3) for k, v in d["stage_seconds"].items()})
PY
done
