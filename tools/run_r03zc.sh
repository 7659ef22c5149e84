cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zc; mkdir -p $out
timeout -k 10 500 python3 tools/divergence_probe.py > $out/divergence.txt 2>&1 || { tail -20 $out/divergence.txt; exit 1; }
grep -v amdgpu.ids $out/divergence.txt
