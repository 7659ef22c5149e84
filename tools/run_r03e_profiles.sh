# final profile set of round 3 (run through gpurun): the shipped sources
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03e || echo "r03e failed"
bash tools/profile_round.sh r03e_c1 --config 1 || echo "r03e_c1 failed"
bash tools/profile_round.sh r03e_af --config 1 --array-forcing --ring-days 10 || echo "af failed"
bash tools/profile_round.sh r03e_coarse --config 1 --coarse 8x8 --ring-days 10 || echo "coarse failed"
