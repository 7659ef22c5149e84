#!/bin/bash
# round 5, call V: the whole GPU suite and smoke() on the final tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05v; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1
rc=$?; tail -4 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
