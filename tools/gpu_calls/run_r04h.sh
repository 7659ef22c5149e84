#!/bin/bash
# round 4, call H: counters of the configs[4] pipeline's kernels as the pipeline runs them
cd "$GRAFT_REPO_ROOT"
bash tools/profile_aux.sh r04_c4 bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify
