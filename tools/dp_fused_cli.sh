#!/bin/bash
# the command line on 10 M pairs with the traceback as its own launch / fused into the paired fill kernel (run through gpurun from the repository root)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
W=/tmp/e2e_w; rm -rf $W; mkdir -p gpurun_out/r03r
python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 --workdir $W > gpurun_out/r03r/e2e_pairs.json 2> gpurun_out/r03r/e2e_pairs.log
for v in 0 1 0 1; do
  rm -f $W/out.gout.1 $W/out.dpout.1 $W/out.unpair
  S3DP_DP_FUSED=$v soap3-dp_amd/soap3-dp pair $W/g.index $W/reads_1.fq $W/reads_2.fq -L 150 -u 500 -v 1 -o $W/out 2>&1 | grep "Alignment + output\|Device time" | sed "s/^/fused=$v: /"
done
