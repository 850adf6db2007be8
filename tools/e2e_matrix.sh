#!/bin/bash
# single reads through the command line on one set of files: stream depth x hardware queues, two runs each (run through gpurun from the repository root)
# tools/e2e_matrix.sh <genome_bp> <repeat_frac>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
G=${1:-3100000000}; R=${2:-0.5}
O=gpurun_out/r03k; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp $G --reads 20000000 --repeat-frac $R --workdir $W > $O/e2e_base_$R.json 2> $O/e2e_base_$R.log
run() {
  local tag=$1; shift
  for k in 1 2; do
    rm -f $W/out.gout.1
    env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out 2>&1 | grep "Alignment + output\|Device time\|Stream loop" | sed "s/^/$tag run $k: /"
  done
}
if [ "$3" = "writers" ]; then
  for t in 8 16 4 8 16; do run writers$t S3DP_WRITE_THREADS=$t; done
  exit 0
fi
if [ "$3" = "buffers" ]; then
  run default A=1
  run big S3DP_REC_PER_READ=64 S3DP_OCC_PER_READ=48
  run pinned40 S3DP_STREAM_OCC_PER_READ=40
  run pinned40_big S3DP_STREAM_OCC_PER_READ=40 S3DP_REC_PER_READ=64 S3DP_OCC_PER_READ=48
  exit 0
fi
if [ -n "$3" ]; then
  for t in $3; do run threads$t S3DP_HOST_THREADS=$t; done
  exit 0
fi
run d3_q4 S3DP_STREAM_DEPTH=3 GPU_MAX_HW_QUEUES=4
run d3_q16 S3DP_STREAM_DEPTH=3 GPU_MAX_HW_QUEUES=16
run d4_q16 S3DP_STREAM_DEPTH=4 GPU_MAX_HW_QUEUES=16
run d6_q16 S3DP_STREAM_DEPTH=6 GPU_MAX_HW_QUEUES=16
run d6_q4 S3DP_STREAM_DEPTH=6 GPU_MAX_HW_QUEUES=4
