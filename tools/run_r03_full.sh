cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_full; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
