# final profile set of round 3 (run through gpurun): the shipped sources
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03d || echo "r03d failed"
bash tools/profile_round.sh r03d_c1 --config 1 || echo "r03d_c1 failed"
bash tools/profile_round.sh r03d_af --config 1 --array-forcing --ring-days 10 || echo "af failed"
bash tools/profile_round.sh r03d_coarse --config 1 --coarse 8x8 --ring-days 10 || echo "coarse failed"
