#!/bin/bash
# round 4: the single-end command line (FASTQ in, SAM out) on the repeat-rich 3.1 Gbp genome under stream depth / chunk size settings, one set of files
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
S3DP_VERBOSE=1 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 --workdir $W > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; cut -c1-700 $O/e2e_se_repeat.json; echo
run() {
  local tag=$1; shift
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out 2>&1 | grep "Stream loop\|Alignment + output\|Read load" | sed "s/^/$tag: /"
}
run base A=1
run depth4 S3DP_STREAM_DEPTH=4
run depth6 S3DP_STREAM_DEPTH=6
run chunk2m S3DP_CHUNK_READS=2097152
run chunk4m S3DP_CHUNK_READS=4194304
run chunk2m_d4 S3DP_CHUNK_READS=2097152 S3DP_STREAM_DEPTH=4
run chunk4m_d2 S3DP_CHUNK_READS=4194304 S3DP_STREAM_DEPTH=2
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
run devnull A=1
run devnull_d6 S3DP_STREAM_DEPTH=6
rm -f $W/out.gout.1
