set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03o; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_snow_micro_pipeline_gpu.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
