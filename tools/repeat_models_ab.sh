#!/bin/bash
# round 5: settings on the other repeat models (own genome and index each); run through gpurun:  [CFGS="cfg cfg ..."] tools/repeat_models_ab.sh <outdir> model [model ...]
# (cfg: VAR=val,VAR=val | base; default: the two orderings of the listed reads)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05m}; shift; mkdir -p $O
Q="--uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0"
for m in "$@"; do
  mkdir -p /tmp/ix_$m
  timeout 500 python3 bench.py --repeat-model $m --index-dir /tmp/ix_$m --build-only 1 > /dev/null 2> $O/index_$m.log; echo "index $m rc=$?"
  for cfg in ${CFGS:-S3DP_SORT_LISTED=0 base}; do
    envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
    env $envs timeout 300 python3 bench.py $Q --repeat-model $m --index-dir /tmp/ix_$m --steps 3 --warmup 1 > $O/${m}_$(echo $cfg | tr ',=' '__').json 2> $O/${m}_$(echo $cfg | tr ',=' '__').log; echo "$m $cfg rc=$?"
    grep "resident:" $O/${m}_$(echo $cfg | tr ',=' '__').log
  done
  rm -rf /tmp/ix_$m
done
