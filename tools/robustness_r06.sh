#!/bin/bash
# round 6: the other repeat models of bench.py at the headline's size (10 M x 100 bp, own genome and index each): tools/robustness_r06.sh <outdir> model [model ...]
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06rob}; shift; mkdir -p $O
Q="--uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 2048 --exact-rank 0 --robustness= --cpu-reference 0"
for m in "$@"; do
  rm -rf /tmp/s3dp_ix_$m; mkdir -p /tmp/s3dp_ix_$m
  timeout 900 python3 bench.py --repeat-model $m --index-dir /tmp/s3dp_ix_$m $Q --steps 4 --warmup 1 > $O/$m.json 2> $O/$m.log; echo "$m rc=$?"
  grep "resident:" $O/$m.log | cut -c1-220
  python3 - $O/$m.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); pk = d["roofline"]["per_kernel"]
    print("   per kernel ms:", {k: round(v["ms_per_step"], 1) for k, v in pk.items()}, "oracle", (d.get("parity") or {}).get("oracle_check_bit_exact"))
except Exception as e:
    print("   (no json)", e)
PY
  rm -rf /tmp/s3dp_ix_$m
done
