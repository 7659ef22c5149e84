#!/bin/bash
# round 5, call G: the whole GPU suite on the round's tree (what the driver runs at round end), smoke
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05g; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1
rc=$?; tail -8 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $o/smoke.txt
