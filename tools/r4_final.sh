#!/bin/bash
# round-4 measurement set (run through gpurun from the repository root): kernel statistics and counters of the bench ladder on the repeat-rich genome
# (profiles/pmc_traffic.json's source), the paired BWT stage under rocprofv3, the command line on 10 M C3 pairs under rocprofv3.  The default
# `python bench.py` line is tools/r4_bench.sh.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export S3DP_PROFILE_OUT=gpurun_out/r04
O=$S3DP_PROFILE_OUT
mkdir -p $O
tools/run_profiles.sh stats pmc 2>&1 | tail -30
python3 tools/make_pmc_traffic.py "$(git rev-parse HEAD 2>/dev/null || cat .commit 2>/dev/null || echo unknown)" 0.5=/tmp/pmc_fetch_0.5.csv > $O/pmc_traffic.json; head -c 700 $O/pmc_traffic.json; echo
# the paired BWT stage (2 M pairs, resident)
rm -rf /tmp/prof_pairs
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pairs -o pairs -- python3 tools/bench_pairs.py --counters 0 --index-dir /tmp/s3dp_ix > $O/pairs_bwt_stage.json 2> $O/pairs.log; echo "pairs rc=$?"
f=$(find /tmp/prof_pairs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_pairs.csv && grep "s3::\|rocprim" $O/kernel_stats_pairs.csv | head -14 | cut -c1-150
cut -c1-600 $O/pairs_bwt_stage.json
du -sh $O
