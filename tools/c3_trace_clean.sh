#!/bin/bash
# round 6: the timeline of the paired command line as the FIRST run on fresh files (a run that overwrites the previous run's 7 GB of SAM files, or starts while they are
# still being written back, is a second slower: tools/c3_ab.sh) -- tools/bench_c3.py (its own run of the command line first), the outputs deleted and synced, then the command line under
# rocprofv3 --kernel-trace --hip-trace, tools/trace_gaps.py and tools/trace_api_stalls.py on the window of the search
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06c3t}; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/c3w /tmp/prof_c3
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3.json | head -1
# the first run's SAM files go, and what is left of them in the page cache is written back, before the profiled run starts
rm -f /tmp/c3w/out*; sync; sleep ${PAUSE:-10}
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/prof_c3 -o cli -- $CLI -o /tmp/c3w/outq > $O/prof.log 2>&1; echo "prof rc=$? $(grep 'Alignment + output' $O/prof.log)"
grep "Device time\|Chunk loop\|DP stages time" $O/prof.log
f=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_gaps.py $f from firstKernel > $O/trace_gaps.txt && head -8 $O/trace_gaps.txt && sed -n '/idle time by the kernel/,/summed kernel time/p' $O/trace_gaps.txt | head -14
a=$(find /tmp/prof_c3 -name "*hip_api_trace.csv" | head -1)
[ -n "$a" ] && python3 tools/trace_api_stalls.py $a $f firstKernel 15 > $O/trace_api_stalls.txt && head -30 $O/trace_api_stalls.txt
rm -rf /tmp/c3w /tmp/prof_c3 /tmp/s3dp_ix
