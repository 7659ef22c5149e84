#!/bin/bash
# round 5 profile set, part b: configs[1], array forcing, coarse forcing, and the configs[4] pipeline's kernels
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05_c1 --config 1 || echo "r05_c1 failed"
bash tools/profile_round.sh r05_af --config 1 --array-forcing --ring-days 10 || echo "af failed"
bash tools/profile_round.sh r05_coarse --config 1 --coarse 8x8 --ring-days 10 || echo "coarse failed"
bash tools/profile_aux.sh r05_c4 bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify || echo "c4 aux failed"
