#!/bin/bash
# the round's closing measurements on one GPU box: the driver's bench line, kernel statistics of the bench ladder under rocprofv3, FETCH_SIZE of the search
# kernels (-> profiles/pmc_traffic.json).   tools/round_final.sh <outdir> [bench|stats|pmc ...]   (default: all three; "pmc stats bench" makes the bench line carry the counters of the same box and commit)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05z}; shift; mkdir -p $O
WHAT="${*:-bench stats pmc}"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0"
for w in $WHAT; do
  case $w in
  bench)
    rm -rf /tmp/s3dp_ix /tmp/c3w /tmp/e2e_w /tmp/prof_*
    timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
    tail -12 $O/bench.log | cut -c1-400 ;;
  stats)
    mkdir -p /tmp/s3dp_ix
    timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
    rm -rf /tmp/prof_stats
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 bench.py $Q --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log; echo "stats rc=$?"
    f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bench.csv && grep "s3::" $O/kernel_stats_bench.csv | head -14 | cut -c1-150
    t=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/ladder_breakdown.py "$t" > $O/ladder_breakdown.txt 2>&1 && head -30 $O/ladder_breakdown.txt
    rm -rf /tmp/prof_stats ;;
  pmc)
    S3DP_COMMIT=$(cat .commit 2>/dev/null | head -c 12) tools/ladder_pmc.sh $(basename $O)
    # (a bench stage that follows in the same call reports THIS measurement: roofline.traffic, per_kernel.fetched_bytes, traffic_measured_at_commit)
    [ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json ;;
  esac
done
