#!/bin/bash
# round 5: the paired command line on the 10 M C3 pairs under chunk sizes (reads per chunk) and DP thread counts; files of tools/c3_profile.sh reused when present
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05k}; mkdir -p $O /tmp/s3dp_ix
if [ ! -f /tmp/c3w/r1.fq ]; then
  timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
  timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
fi
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
OPT="-L 150 -u 500 -v 1 -b 2"
run() { local tag=$1; shift; env "$@" timeout 200 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/run_$tag.txt 2>&1; echo "$tag rc=$? $(grep 'Alignment + output' $O/run_$tag.txt) | $(grep 'Device time' $O/run_$tag.txt | cut -c1-120)"; }
rm -rf /tmp/e2e_w /tmp/prof_*
# (the three SAM files are links to /dev/null: a run that overwrites 8 GB of files still being written back takes up to two seconds longer, which would be
#  what the sweep measures; `first` below is the one run with real files)
run first A=1
rm -f /tmp/c3w/outp.*
for f in gout.1 dpout.1 unpair; do ln -s /dev/null /tmp/c3w/outp.$f; done
run c1m A=1
run c1m_prio0 S3DP_DP_PRIORITY=0
run c2m S3DP_CHUNK_READS=2097152
run c2m_prio0 S3DP_CHUNK_READS=2097152 S3DP_DP_PRIORITY=0
run c4m S3DP_CHUNK_READS=4194304
run c1m_b3 S3DP_BLOCKS_PER_CU=3
run c2m_b3 S3DP_CHUNK_READS=2097152 S3DP_BLOCKS_PER_CU=3
run c1m_1dp S3DP_DP_THREADS=1
run c2m_again S3DP_CHUNK_READS=2097152
run c1m_again A=1
rm -rf /tmp/c3w/outp*
