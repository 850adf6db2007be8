#!/bin/bash
# the thresholds of the heavy-read machinery on other repeat models (bench.py --repeat-model): the defaults against round 2's settings and a third point;
# one index build per model (kept in /tmp), the resident ladder only.  Output: one line per (model, setting).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/ixr
for model in young old satellite-heavy; do
  Q="--index-dir /tmp/ixr --repeat-model $model --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 512 --exact-rank 0 --steps 2 --warmup 1"
  timeout 600 python3 bench.py --index-dir /tmp/ixr --repeat-model $model --build-only 1 > /dev/null 2>&1
  for cfg in "S3DP_NOOP=1" "S3DP_HEAVY=1024 S3DP_HEAVY_RANK=4096 S3DP_CE_GIVE_UP=4" "S3DP_HEAVY=4096 S3DP_HEAVY_RANK=2048 S3DP_CE_GIVE_UP=2" "S3DP_HEAVY=1024 S3DP_HEAVY_RANK=512 S3DP_CE_GIVE_UP=2"; do
    env $cfg timeout 400 python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$model | $cfg | ms_per_step %.1f  M reads/s %.2f  stepping %.1f  ce %.1f  rank/read %.0f  truth %s  oracle %s' % (d['ms_per_step'], d['value'], r['stepping_kernel_ms_per_step'], r['check_extend_kernel_ms_per_step'], d['rank_queries_per_read'], d['truth_check']['ok'], d.get('oracle_check',{}).get('bit_exact')))
"
  done
  rm -f /tmp/ixr/*
done
