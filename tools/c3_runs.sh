#!/bin/bash
# the paired command line on the 10 M C3 pairs (files of tools/c3_profile.sh / c3_runs.sh reused when present) under environment settings, outputs to /dev/null:
#   tools/c3_runs.sh <outdir> tag[:VAR=val,VAR=val] ...         (every run twice; the lines give "Alignment + output" and the device times)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05c3}; shift; mkdir -p $O /tmp/s3dp_ix
if [ ! -f /tmp/c3w/r1.fq ]; then
  timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
  timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
fi
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
OPT="-L 150 -u 500 -v 1 -b 2"
rm -rf /tmp/e2e_w /tmp/prof_* /tmp/c3w/outp*
for f in gout.1 dpout.1 unpair; do ln -s /dev/null /tmp/c3w/outp.$f; done
for spec in "$@"; do
  tag=${spec%%:*}; envs=""; [ "$spec" != "$tag" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  for rep in a b; do
    env $envs timeout 200 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/run_${tag}_$rep.txt 2>&1
    echo "$tag $rep rc=$? $(grep 'Alignment + output' $O/run_${tag}_$rep.txt) | $(grep 'DP stages time' $O/run_${tag}_$rep.txt) | $(grep 'Device time' $O/run_${tag}_$rep.txt | cut -c1-140)"
  done
done
rm -rf /tmp/c3w/outp*
