#!/bin/bash
# round 6: device allocations and buffer growths of 16 MB and more in the paired command line, with their durations (S3DP_VERBOSE=1), as a clean run (outputs of the
# run before deleted and synced): tools/c3_allocs.sh <outdir> [base | VAR=val,... ...]
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06c3a}; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/c3w
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3.json | head -1
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
shift
[ $# -eq 0 ] && set -- base
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ',=' '__')
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  for k in 1 2; do
    rm -f /tmp/c3w/out*; sync; sleep ${PAUSE:-8}
    env $envs S3DP_VERBOSE=1 timeout 200 $CLI -o /tmp/c3w/outv > $O/verbose_${name}_$k.log 2>&1; echo "$cfg run $k rc=$? $(grep 'Alignment + output' $O/verbose_${name}_$k.log)"
  done
  grep "\[s3dp\]" $O/verbose_${name}_2.log | grep " MB " | awk '$(NF-1) >= 5.0' | cut -c1-160 | head -30
  echo "$cfg: allocations and growths of 1 MB and more: $(grep '\[s3dp\]' $O/verbose_${name}_2.log | grep -c ' MB '), $(grep '\[s3dp\]' $O/verbose_${name}_2.log | grep ' MB ' | awk '{s+=$(NF-1)} END {print s}') ms"
done
rm -rf /tmp/c3w /tmp/s3dp_ix
