#!/bin/bash
# round 4, call E: the whole GPU suite on the tree (ADVICE fixes, 42-cell pruning, new tests)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04e; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; rc=$?
tail -8 $o/tests.log
exit $rc
