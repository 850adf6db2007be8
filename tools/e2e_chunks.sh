#!/bin/bash
# the command line under several chunk sizes on the repeat-rich genome (single reads) and on pairs (run through gpurun from the repository root)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 --workdir $W > $O/e2e_base.json 2> $O/e2e_base.log; cut -c1-700 $O/e2e_base.json
run() {
  local tag=$1; shift
  rm -f $W/out.gout.1
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out 2>&1 | grep "Stream loop\|Alignment + output" | sed "s/^/$tag: /"
}
run c1m A=1
run c2m S3DP_CHUNK_READS=2097152
run c4m S3DP_CHUNK_READS=4194304
run c1m_b4 S3DP_STREAM_BLOCKS_PER_CU=4
run c1m_b1 S3DP_STREAM_BLOCKS_PER_CU=1
run c2m_b4 S3DP_CHUNK_READS=2097152 S3DP_STREAM_BLOCKS_PER_CU=4
rm -rf $W
python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 --workdir $W > $O/e2e_pairs.json 2> $O/e2e_pairs.log; cut -c1-500 $O/e2e_pairs.json
runp() {
  local tag=$1; shift
  rm -f $W/out.gout.1 $W/out.dpout.1 $W/out.unpair
  env "$@" soap3-dp_amd/soap3-dp pair $W/g.index $W/reads_1.fq $W/reads_2.fq -L 150 -u 500 -v 1 -o $W/out 2>&1 | grep "Chunk loop\|Alignment + output\|DP stages time" | sed "s/^/$tag: /"
}
runp p1m A=1
runp p2m S3DP_CHUNK_READS=2097152
runp p4m S3DP_CHUNK_READS=4194304
