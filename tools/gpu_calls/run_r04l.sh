#!/bin/bash
# round 4, call L: configs[4] share on the final tree, its stage seconds, and what a 167 GB hipMalloc costs (the bioclim question)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04l; mkdir -p $o
timeout -k 10 1000 python bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_c4.json 2> $o/bench_c4.err || { tail -20 $o/bench_c4.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04l/bench_c4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("stage_seconds") or d["config"].get("stage_seconds"), indent=0))
print(json.dumps(d.get("verified"))[:1500])
PY
python - <<'PY'
import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (10, 40, 167):
    t = time.perf_counter(); x = torch.empty(gb * 10**9, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); a = time.perf_counter() - t
    t = time.perf_counter(); del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); f = time.perf_counter() - t
    print(f"hipMalloc {gb} GB: {a:.3f} s, free {f:.3f} s")
PY
