#!/bin/bash
# round 4, end of round: the GPU suite and smoke() at HEAD
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04y2; mkdir -p $O
timeout 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
