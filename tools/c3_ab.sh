#!/bin/bash
# round 6: the paired command line (BASELINE.json configs[2]: 2 x 150 bp, DP stages on; PAIRS pairs of the benchmark genome) under environment settings:
# tools/c3_ab.sh <outdir> cfg [cfg ...]   (cfg: VAR=val,VAR=val | base); two runs per setting, the command line's own times
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06c3ab}; shift; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/c3w
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 600 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3.json | head -2
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ',=' '__')
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  for k in 1 2; do
    sleep ${PAUSE:-6}      # (a process that starts while the last one's 219 GB are still being given back pays for it: tools/exit_fast_probe.sh)
    env $envs timeout 200 $CLI -o /tmp/c3w/outp > $O/${name}_$k.log 2>&1
    echo "$cfg run $k rc=$? $(grep 'Alignment + output' $O/${name}_$k.log) | $(grep 'Device time' $O/${name}_$k.log | head -1 | cut -c1-160)"
  done
done
rm -rf /tmp/c3w
