#!/bin/bash
# round 6: the timeline of the paired command line (BASELINE.json configs[2]: 2 x 150 bp, DP stages on; PAIRS pairs of the benchmark genome, outputs kept in
# /tmp): two plain runs, then one under rocprofv3 --kernel-trace --hip-trace with tools/trace_gaps.py and tools/trace_api_stalls.py on the window of the search
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06c3}; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/c3w /tmp/prof_c3
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-5000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3.json | head -2
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
for k in 1 2; do timeout 120 $CLI -o /tmp/c3w/outp > $O/plain_$k.log 2>&1; echo "plain $k rc=$? $(grep 'Alignment + output' $O/plain_$k.log)"; done
grep "Device time\|Chunk loop\|DP stages time" $O/plain_2.log
timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/prof_c3 -o cli -- $CLI -o /tmp/c3w/outq > $O/prof.log 2>&1; echo "prof rc=$? $(grep 'Alignment + output' $O/prof.log)"
f=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_gaps.py $f from firstKernel | tee $O/trace_gaps.txt
a=$(find /tmp/prof_c3 -name "*hip_api_trace.csv" | head -1)
[ -n "$a" ] && python3 tools/trace_api_stalls.py $a $f firstKernel 15 | tee $O/trace_api_stalls.txt
rm -rf /tmp/c3w /tmp/prof_c3 /tmp/s3dp_ix
