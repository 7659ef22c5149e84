#!/bin/bash
# round 5 profile set, part c: the configs[4] pipeline's kernels again (after the snow-day microclimate kernel's last change)
cd $GRAFT_REPO_ROOT
bash tools/profile_aux.sh r05_c4 bench.py --config 4 --share 8 --steps 1 --warmup 0 --no-cpu-baseline --no-verify || echo "c4 aux failed"
