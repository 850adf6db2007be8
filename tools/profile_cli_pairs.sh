#!/bin/bash
# the command line on 10 M 2 x 150 bp pairs (DP stages on) under rocprofv3 --kernel-trace --stats (run through gpurun from the repository root)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03q; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 --workdir $W > $O/e2e_pairs.json 2> $O/e2e_pairs.log
rm -f $W/out.gout.1 $W/out.dpout.1 $W/out.unpair; rm -rf /tmp/prof_cli
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli -o cli -- soap3-dp_amd/soap3-dp pair $W/g.index $W/reads_1.fq $W/reads_2.fq -L 150 -u 500 -v 1 -o $W/out > $O/cli_under_rocprof.log 2>&1
grep "Alignment + output\|Device time\|Chunk loop" $O/cli_under_rocprof.log
f=$(find /tmp/prof_cli -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_cli_pairs.csv && head -24 $O/kernel_stats_cli_pairs.csv | cut -c1-90,150-260
