set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03k; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
CONFIG=2 tools/ab_bench2.sh $out/ab2 r02=build/variants/libmcfhip_r02.so new=-
bash tools/profile_round.sh r03a
