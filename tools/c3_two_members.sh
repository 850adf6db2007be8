#!/bin/bash
# round 5 experiment: the paired command line with TWO members on one device (-c 0,0: two chunk loops, two sets of DP stage threads) against one
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05m2}; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/e2e_w /tmp/prof_*
if [ ! -f /tmp/c3w/r1.fq ]; then
  timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
  timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
fi
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
OPT="-L 150 -u 500 -v 1 -b 2"
rm -f /tmp/c3w/outp.*; for f in gout.1 dpout.1 unpair gout.2 dpout.2 unpair.2; do ln -s /dev/null /tmp/c3w/outp.$f; done
run() { local tag=$1; shift; local dev=$1; shift; env "$@" timeout 200 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -c $dev -o /tmp/c3w/outp > $O/run_$tag.txt 2>&1
        echo "$tag rc=$? $(grep 'Alignment + output' $O/run_$tag.txt | tr -s ' ') | $(grep 'Device time' $O/run_$tag.txt | cut -c8-100) | $(grep 'Lookup tables' $O/run_$tag.txt | tr '\n' ' ')"; }
for rep in 1 2; do
  run one_r$rep 0 A=1
  run two_r$rep 0,0 A=1
  run two_dp1_r$rep 0,0 S3DP_DP_THREADS=1
done
