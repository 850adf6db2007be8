#!/bin/bash
# round 4: single-end 150 bp reads in the default mode (DP on, 100-base seeds verified on the device) through the command line under rocprofv3: which kernel takes the time?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 400 python3 tools/bench_c3.py --pairs 1000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_1m.json 2> $O/c3_1m.log; echo "c3 rc=$?"
SE="soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -b 2"
export S3DP_VERBOSE=1
timeout 200 $SE -o /tmp/c3w/se_plain > $O/se_plain.log 2>&1; echo "plain rc=$? $(grep 'Alignment + output\|Device time' $O/se_plain.log)"
timeout 200 $SE -s 2 -o /tmp/c3w/se_nodp > $O/se_nodp.log 2>&1; echo "DP off (-s 2) rc=$? $(grep 'Alignment + output\|Device time' $O/se_nodp.log)"
rm -rf /tmp/prof_se
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_se -o se -- $SE -o /tmp/c3w/se_prof > $O/se_prof.log 2>&1; echo "prof rc=$?"
f=$(find /tmp/prof_se -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/se150_dp_kernel_stats.csv && head -12 $O/se150_dp_kernel_stats.csv | cut -c1-200
