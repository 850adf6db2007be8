#!/bin/bash
# round 5: the fast trips of the unit pass (S3DP_FAST_TRIPS / S3DP_FAST_MIN) and of the first / trunk pass (S3DP_FIRST_FAST_TRIPS / S3DP_FIRST_FAST_MIN):
# parity with the fast path forced on the fixtures (any stepping lane makes a fast trip), then the bench ladder per setting.
# tools/fast_ab.sh <outdir> cfg [cfg ...]      (cfg as in tools/ladder_ab.sh; PARITY=0: ladder only)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05f}; mkdir -p $O
if [ "${PARITY:-1}" = 1 ]; then
  S3DP_FAST_MIN=1 S3DP_FAST_TRIPS=${PARITY_TRIPS:-3} S3DP_FIRST_FAST_MIN=1 S3DP_FIRST_FAST_TRIPS=${PARITY_TRIPS:-3} timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_sampled_sa.py -m gpu -x -q > $O/pytest_fast_forced.log 2>&1; echo "parity (fast trips forced) rc=$?"; tail -2 $O/pytest_fast_forced.log
fi
exec tools/ladder_ab.sh "$@"
