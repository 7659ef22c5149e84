set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03q; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_multi_device_gpu.py tests/test_parity_gpu.py tests/test_abi_cpu.py -x -q > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
