#!/bin/bash
# round 5: `single -L 150 -h 1` (all valid alignments of long reads: 100-base seeds searched without a hit limit) on the first mates of the 10 M C3 pairs,
# repeat-rich 3.1 Gbp genome -- the mode that ran out of HBM at 1 Mi reads per chunk in round 4; now decoded in slices (S3DP_LONG_BUDGET)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05h}; mkdir -p $O /tmp/s3dp_ix
rm -rf /tmp/e2e_w /tmp/prof_*
if [ ! -f /tmp/c3w/r1.fq ]; then
  timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
  timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 files rc=$?"
fi
ls -la /tmp/c3w | head; ln -sf /dev/null /tmp/c3w/h1.gout.1; ln -sf /dev/null /tmp/c3w/h1.dpout.1
run() { local tag=$1; shift; env "$@" timeout 300 soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -h 1 -b 2 -o /tmp/c3w/h1 > $O/h1_$tag.txt 2>&1
        echo "$tag rc=$? $(grep 'Alignment + output' $O/h1_$tag.txt | tr -s ' ') | $(grep 'Device time' $O/h1_$tag.txt | cut -c8-90) | $(tail -1 $O/h1_$tag.txt | cut -c1-150)"; }
run chunk256k A=1
run chunk1m S3DP_CHUNK_READS=1048576
run chunk1m_budget26 S3DP_CHUNK_READS=1048576 S3DP_LONG_BUDGET=67108864
run chunk2m S3DP_CHUNK_READS=2097152
