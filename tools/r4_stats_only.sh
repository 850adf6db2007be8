cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0"
timeout 200 python3 bench.py $Q --steps 3 --warmup 1 > $O/plain.json 2> $O/plain.log; echo "plain rc=$?"; cut -c1-200 $O/plain.json
rm -rf /tmp/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 bench.py $Q --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log; echo "stats rc=$?"
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bench.csv && grep "s3::" $O/kernel_stats_bench.csv | head -14 | cut -c1-150
tail -3 $O/bench_under_rocprof.log | cut -c1-200
