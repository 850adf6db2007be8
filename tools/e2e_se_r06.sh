#!/bin/bash
# round 6: the single-end command line (FASTQ in, SAM to /dev/null) on the repeat-rich 3.1 Gbp genome, READS reads (default 40 M: a run of 38 chunks is
# no longer mostly fill and drain), 1 Mi-read chunks x 3 / 6 slots and 2 Mi x 3, once more with the program's timeline (S3DP_VERBOSE=1)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06s}; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W /tmp/c3w /tmp/s3dp_ix /tmp/prof_*
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads ${READS:-40000000} --repeat-frac 0.5 --workdir $W > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; cut -c1-400 $O/e2e_se_repeat.json; echo
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
run() {
  local tag=$1; shift
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out > $O/run_$tag.txt 2>&1
  grep "Stream loop\|Alignment + output\|Device time\|Flank\|feeder\|waited" $O/run_$tag.txt | sed "s/^/$tag: /" | cut -c1-300
}
run c1m_d3 S3DP_CHUNK_READS=1048576 S3DP_STREAM_DEPTH=3
run c1m_d6 S3DP_CHUNK_READS=1048576 S3DP_STREAM_DEPTH=6
run c2m_d3 S3DP_CHUNK_READS=2097152 S3DP_STREAM_DEPTH=3
run c2m_d4 S3DP_CHUNK_READS=2097152 S3DP_STREAM_DEPTH=4
run c1m_d6_verbose S3DP_CHUNK_READS=1048576 S3DP_STREAM_DEPTH=6 S3DP_VERBOSE=1
tail -40 $O/run_c1m_d6_verbose.txt | cut -c1-300
rm -rf $W
