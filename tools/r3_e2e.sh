#!/bin/bash
# the command line end to end, four workloads (run through gpurun from the repository root); results under gpurun_out/$1 (default r03c)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03c}; mkdir -p $O
S3DP_VERBOSE=1 timeout 900 python3 tools/bench_e2e.py --genome-bp 1000000000 --reads 20000000 > $O/e2e_se_uniform_1g.json 2> $O/e2e_se_uniform_1g.log; echo "1g rc=$?"; cut -c1-800 $O/e2e_se_uniform_1g.json
S3DP_VERBOSE=1 timeout 900 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 > $O/e2e_se_uniform.json 2> $O/e2e_se_uniform.log; echo "uniform rc=$?"; cut -c1-800 $O/e2e_se_uniform.json
S3DP_VERBOSE=1 timeout 1200 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; echo "repeat rc=$?"; cut -c1-800 $O/e2e_se_repeat.json
S3DP_VERBOSE=1 timeout 1200 python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 > $O/e2e_pairs_dp.json 2> $O/e2e_pairs_dp.log; echo "pairs rc=$?"; cut -c1-1000 $O/e2e_pairs_dp.json
grep -h "s3dp\] member" $O/e2e_pairs_dp.log | tail -12 | cut -c1-200
