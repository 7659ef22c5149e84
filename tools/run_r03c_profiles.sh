# final profile set of round 3 (run through gpurun): the shipped sources
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03c || echo "r03c failed"
bash tools/profile_round.sh r03c_c1 --config 1 || echo "r03c_c1 failed"
bash tools/profile_round.sh r03c_af --config 1 --array-forcing --ring-days 5 || echo "af failed"
bash tools/profile_round.sh r03c_coarse --config 1 --coarse 8x8 --ring-days 5 || echo "coarse failed"
