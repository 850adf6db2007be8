#!/bin/bash
# round 5: the single-end command line (FASTQ in, SAM out) on the repeat-rich 3.1 Gbp genome: chunk size x stream depth, results kept in files
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05e}; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W /tmp/c3w /tmp/s3dp_ix /tmp/prof_*          # (the boxes' /tmp holds about two sets of index files and one run's output)
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 --workdir $W > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; cut -c1-600 $O/e2e_se_repeat.json; echo
run() {
  local tag=$1; shift
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out > $O/run_$tag.txt 2>&1
  grep "Stream loop\|Alignment + output\|Device time" $O/run_$tag.txt | sed "s/^/$tag: /"
}
if [ "${DISK:-1}" = 1 ]; then
  for c in 1048576 2097152 4194304; do for d in 2 3 4; do run c${c}_d$d S3DP_CHUNK_READS=$c S3DP_STREAM_DEPTH=$d; done; done
  run default A=1
fi
# the same with the SAM file a link to /dev/null: what the program does without the box's disk (3.3 GB/s here: 7.9 GB of SAM for 20 M reads)
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
for c in 1048576 2097152 4194304; do for d in 2 3 4; do run devnull_c${c}_d$d S3DP_CHUNK_READS=$c S3DP_STREAM_DEPTH=$d; done; done
run devnull_default A=1
rm -f $W/out.gout.1
rm -rf $W
