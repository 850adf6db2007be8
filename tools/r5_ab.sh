#!/bin/bash
# round 5: the ladder with and without the two orderings (listed reads / the batch), per-launch statistics of the counted pass; run through gpurun
#   tools/r5_ab.sh <outdir> "<env settings A>" "<env settings B>" ...      (each setting list: VAR=val,VAR=val or "base")
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05a}; shift
mkdir -p $O /tmp/s3dp_ix
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check ${CHECK:-0} --exact-rank 0 --robustness= --cpu-reference 0"
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ',=' '__')
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  env $envs S3DP_VERBOSE=1 timeout 300 python3 bench.py $Q --steps ${STEPS:-3} --warmup 1 > $O/$name.json 2> $O/$name.log; echo "$cfg rc=$?"
  grep "^\[s3dp\]" $O/$name.log | cut -c1-420 | head -60
  grep "resident:" $O/$name.log
done
