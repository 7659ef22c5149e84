#!/bin/bash
# round 4 profile set, part a: the default bench (configs[2]) — final kernel sources
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04 || echo "r04 failed"
