#!/bin/bash
# round 4, end of round: the paired BWT stage (resident, no counted pass) and the command line on 10 M C3 pairs under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04r; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
rm -rf /tmp/prof_pairs
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pairs -o pairs -- python3 tools/bench_pairs.py --counters 0 --index-dir /tmp/s3dp_ix > $O/pairs_bwt_stage.json 2> $O/pairs.log; echo "pairs rc=$?"
f=$(find /tmp/prof_pairs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_pairs.csv && grep "s3::\|rocprim" $O/kernel_stats_pairs.csv | head -12 | cut -c1-140
cut -c1-400 $O/pairs_bwt_stage.json
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"; grep -o "\"end_to_end.*lookup_chars\": 16}" $O/c3_10m.json | cut -c1-900
rm -rf /tmp/prof_cli
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli -o cli -- soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2 -o /tmp/c3w/outp > $O/cli_under_rocprof.log 2>&1; echo "cli rc=$?"
f=$(find /tmp/prof_cli -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cli_pairs_kernel_stats.csv && head -8 $O/cli_pairs_kernel_stats.csv | cut -c1-140
grep "Alignment + output\|Device time" $O/cli_under_rocprof.log
