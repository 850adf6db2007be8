#!/bin/bash
# the paired BWT stage under rocprofv3 (kernel stats), then the bench with its extras (C3 leg, 150 bp single-end leg); index files kept in /tmp between the two
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=${S3DP_PROFILE_OUT:-gpurun_out/r03p}; mkdir -p $OUT /tmp/ix
timeout 600 python3 bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2> $OUT/index_build.log
rm -rf /tmp/prof_pairs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pairs -o pairs -- python3 tools/bench_pairs.py --index-dir /tmp/ix > $OUT/pairs.json 2> $OUT/pairs.log
echo "pairs rc=$?"; cat $OUT/pairs.json | cut -c1-400
f=$(find /tmp/prof_pairs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_pairs.csv && head -14 $OUT/kernel_stats_pairs.csv | cut -c1-70,150-230
timeout 900 python3 bench.py --index-dir /tmp/ix --steps 3 --uniform 0 --streaming 0 --cpu-budget 0 > $OUT/bench_extras.json 2> $OUT/bench_extras.log
echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/bench_extras.json')); print(d['ms_per_step']); print(json.dumps(d.get('extras'),indent=1)[:6000])"
