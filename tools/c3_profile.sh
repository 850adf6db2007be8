#!/bin/bash
# round 5: the paired command line on the 10 M C3 pairs -- plain timings, then under rocprofv3 --kernel-trace --stats with the stages of the chunks
# SERIALISED (S3DP_SERIAL=1 S3DP_DP_THREADS=1: one host thread's kernels on the device at a time), so that the kernel table shows own durations
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05c}; mkdir -p $O /tmp/s3dp_ix
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3.json
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
OPT="-L 150 -u 500 -v 1 -b 2"
for k in 1 2; do timeout 120 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/plain_$k.log 2>&1; echo "plain $k rc=$? $(grep 'Alignment + output' $O/plain_$k.log)"; done
grep "Device time\|Chunk loop\|DP stages time" $O/plain_2.log
S3DP_SERIAL=1 S3DP_DP_THREADS=1 timeout 200 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/serial.log 2>&1; echo "serial rc=$? $(grep 'Alignment + output' $O/serial.log)"
grep "Device time\|Chunk loop\|DP stages time" $O/serial.log
export S3DP_SERIAL=1 S3DP_DP_THREADS=1
rm -rf /tmp/prof_serial
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o cli -- $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outq > $O/prof_serial.log 2>&1; echo "prof serial rc=$? $(grep 'Alignment + output' $O/prof_serial.log)"
f=$(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_cli_pairs_serial.csv && head -30 $O/kernel_stats_cli_pairs_serial.csv | cut -c1-150
