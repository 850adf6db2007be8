#!/bin/bash
# round 6: the single-end DP parity leg of bench.py at the benchmark genome's size (2 M pairs instead of 10 M), and the copy probe's shapes
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06w}; mkdir -p $O
S3DP_PROBE_COPY_VERBOSE=1 timeout 1500 python3 bench.py --steps 3 --warmup 1 --uniform 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0 \
  --c3-pairs 2000000 --c5-pairs 0 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
grep "copy probe shape" $O/bench.log | sort | uniq -c | sort -k5 | head -20
grep "summary\]" $O/bench.log | cut -c1-600
cp bench_full.json $O/ 2>/dev/null
