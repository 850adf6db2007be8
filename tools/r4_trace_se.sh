#!/bin/bash
# round 4: kernel timeline of the single-end command line on the repeat-rich genome (20 M reads, SAM to /dev/null), and chunk / depth variants
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 40000000 --repeat-frac 0.5 --workdir $W > $O/e2e.json 2> $O/e2e.log; cut -c1-300 $O/e2e.json; echo
ln -sf /dev/null $W/out.gout.1
run() {
  local tag=$1; shift
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out 2>&1 | grep "Stream loop\|Alignment + output\|Device time" | sed "s/^/$tag: /"
}
run base A=1
run chunk2m S3DP_CHUNK_READS=2097152
run chunk4m S3DP_CHUNK_READS=4194304
run chunk4m_d2 S3DP_CHUNK_READS=4194304 S3DP_STREAM_DEPTH=2
run chunk2m_d4 S3DP_CHUNK_READS=2097152 S3DP_STREAM_DEPTH=4
run threads8 S3DP_HOST_THREADS=8
rm -rf /tmp/prof_se
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_se -o se -- soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out > $O/under_rocprof.log 2>&1
grep "Stream loop\|Alignment + output" $O/under_rocprof.log
f=$(find /tmp/prof_se -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f 0.55 0.95 | tee $O/trace_gaps.txt
