#!/bin/bash
# the streamed form at 1 Mi reads per batch under several depths / grid widths (run through gpurun from the repository root)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h; mkdir -p $O /tmp/idx
run() {
  local tag=$1 depth=$2 batch=$3; shift 3
  env "$@" python3 bench.py --steps 2 --uniform 0 --c3-pairs 0 --se150-reads 0 --extras 0 --cpu-budget 0 --exact-rank 0 --check 0 --index-dir /tmp/idx --stream-batch $batch --stream-depth $depth > $O/$tag.json 2> $O/$tag.log
  python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("extras", {}).get("streamed", d.get("streamed"))
print(sys.argv[2], "resident %.1f ms" % d["ms_per_step"], json.dumps(s)[:400])
PY
}
for v in "$@"; do
  IFS=: read tag depth batch blocks <<< "$v"
  run $tag $depth $batch S3DP_STREAM_BLOCKS_PER_CU=$blocks
done
