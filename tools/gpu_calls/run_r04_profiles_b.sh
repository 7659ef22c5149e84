#!/bin/bash
# round 4 profile set, part b: configs[1], array forcing, coarse forcing
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04_c1 --config 1 || echo "r04_c1 failed"
bash tools/profile_round.sh r04_af --config 1 --array-forcing --ring-days 10 || echo "af failed"
bash tools/profile_round.sh r04_coarse --config 1 --coarse 8x8 --ring-days 10 || echo "coarse failed"
