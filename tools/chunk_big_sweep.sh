#!/bin/bash
# round 5: large chunks first (S3DP_CHUNK_BIG) on the paired (10 M C3 pairs, DP on) and the single-end (20 M reads, DP off) command line; outputs are links
# to /dev/null except in the runs marked "files".   tools/chunk_big_sweep.sh <outdir> [pe] [se]
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05b}; shift; mkdir -p $O /tmp/s3dp_ix
WHAT="${*:-pe se}"
for w in $WHAT; do
if [ $w = pe ]; then
  rm -rf /tmp/e2e_w /tmp/prof_*
  timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
  timeout 500 python3 tools/bench_c3.py --pairs ${PAIRS:-10000000} --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
  CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
  OPT="-L 150 -u 500 -v 1 -b 2"
  run() { local tag=$1; shift; env "$@" timeout 200 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/pe_$tag.txt 2>&1
          echo "pe $tag rc=$? $(grep 'Alignment + output' $O/pe_$tag.txt | tr -s ' ') | $(grep 'Device time' $O/pe_$tag.txt | cut -c8-95) | $(grep 'Chunk loop' $O/pe_$tag.txt | cut -c25-120)"; }
  rm -f /tmp/c3w/outp.*; for f in gout.1 dpout.1 unpair; do ln -s /dev/null /tmp/c3w/outp.$f; done
  for rep in 1 2; do
    run big0_r$rep S3DP_CHUNK_BIG=0
    run big2m_dp3_r$rep S3DP_CHUNK_BIG=2097152 S3DP_DP_THREADS=3
    run big4m_dp2_r$rep S3DP_CHUNK_BIG=4194304 S3DP_DP_THREADS=2
    run big4m_dp3_r$rep S3DP_CHUNK_BIG=4194304 S3DP_DP_THREADS=3
    run big4m_dp4_r$rep S3DP_CHUNK_BIG=4194304 S3DP_DP_THREADS=4
    run big8m_dp4_r$rep S3DP_CHUNK_BIG=8388608 S3DP_DP_THREADS=4
  done
  rm -f /tmp/c3w/outp.*
  run files_default A=1
  rm -f /tmp/c3w/outp.*; sync
  run files_big0 S3DP_CHUNK_BIG=0
  rm -rf /tmp/c3w
fi
if [ $w = se ]; then
  W=/tmp/e2e_w; rm -rf $W /tmp/c3w /tmp/s3dp_ix /tmp/prof_*
  python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 --workdir $W > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; cut -c1-300 $O/e2e_se_repeat.json; echo
  runs() { local tag=$1; shift; env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out > $O/se_$tag.txt 2>&1
           echo "se $tag: $(grep 'Alignment + output' $O/se_$tag.txt | tr -s ' ') | $(grep 'Device time' $O/se_$tag.txt | cut -c8-90)"; }
  rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
  for rep in 1 2; do
    runs big0_r$rep S3DP_CHUNK_BIG=0
    runs big2m_r$rep S3DP_CHUNK_BIG=2097152
    runs big4m_r$rep S3DP_CHUNK_BIG=4194304
    runs big4m_d2_r$rep S3DP_CHUNK_BIG=4194304 S3DP_STREAM_DEPTH=2
    runs big8m_d2_r$rep S3DP_CHUNK_BIG=8388608 S3DP_STREAM_DEPTH=2
  done
  rm -f $W/out.gout.1
  runs files_default A=1
  rm -rf $W
fi
done
