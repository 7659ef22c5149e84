#!/bin/bash
# round 5 profile set, part a: the default bench (configs[2]) — rocprofv3 kernel stats of the bench command + counter passes
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 || echo "r05 failed"
