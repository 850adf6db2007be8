#!/bin/bash
# sweep of the heavy-read machinery's knobs on the bench workload (measurement aid): "VAR=value VAR=value" per line of $1, index kept in /tmp
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03t /tmp/ix
Q="--index-dir /tmp/ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --steps 2 --warmup 1"
timeout 600 python3 bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2>&1
while read -r cfg; do
  [ -z "$cfg" ] && continue
  env $cfg timeout 300 python3 bench.py $Q 2> gpurun_out/r03t/log.txt | python3 -c "
import sys, json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$cfg: ms_per_step %.1f stepping %.1f ce %.1f rank/read %.0f ok %s' % (d['ms_per_step'], r['stepping_kernel_ms_per_step'], r['check_extend_kernel_ms_per_step'], d['rank_queries_per_read'], d['truth_check']['ok']))
"
done < "$1"
