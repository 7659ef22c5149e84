#!/bin/bash
# round 4, call P: the configs[4] pipeline's kernels under rocprofv3 on the final tree (stats, SQ counters, HBM bytes)
bash tools/profile_aux.sh r04_c4 bench.py --config 4 --share 8 --steps 1 --warmup 1 --no-cpu-baseline --no-verify
