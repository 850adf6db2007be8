#!/bin/bash
# round 5: unit-pass variants (S3DP_TAKE_MIN ...) on the bench ladder, timed steps only; tools/ladder_ab.sh <outdir> cfg [cfg ...]   (cfg: VAR=val,VAR=val | base)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05u}; shift; mkdir -p $O /tmp/s3dp_ix
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check ${CHECK:-2048} --exact-rank 0 --robustness= --cpu-reference 0"
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ',=' '__')
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  env $envs timeout 300 python3 bench.py $Q --steps ${STEPS:-4} --warmup 1 > $O/$name.json 2> $O/$name.log; echo "$cfg rc=$?"
  grep "resident:" $O/$name.log
  python3 - $O/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    pk = d["roofline"]["per_kernel"]
    print("   per kernel ms:", {k: round(v["ms_per_step"], 1) for k, v in pk.items()}, "oracle", (d.get("parity") or {}).get("oracle_check_bit_exact"))
except Exception as e:
    print("   (no json)", e)
PY
done
