#!/bin/bash
# Regenerates the golden fixtures with the REAL reference (oracle/_ref, built by oracle/ref.mk from
# /root/reference).  Runs only in the authoring container; the outputs below are committed data
# (inputs + expected outputs), never reference source.  index.ann carries a random build stamp: keep the committed one.
#   g1: 200 kbp, 3 chromosomes, 4 N-runs, 6 planted repeat families; 2000 x 100 bp reads (0-4 substitutions)
#   g2: 60 kbp, heavy repeats (exceeds MaxOutputPerRead=1000 with the small-cap ini), mixed read lengths 25..120
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=$ROOT/oracle/_ref
INI=$ROOT/soap3-dp_amd/host/soap3-dp.ini
OUT=${S3_FIXTURE_OUT:-$ROOT/tests/golden}
W=$(mktemp -d)
trap 'rm -rf $W' EXIT

mk() {  # name genome_args read_args maxlen
  local name=$1; shift
  local gargs=$1; shift
  local rargs=$1; shift
  local L=$1; shift
  mkdir -p $OUT/$name
  ( cd $W && python $ROOT/tools/simulate.py $gargs --fasta $name.fa $rargs --fastq $name.fq --truth $name.truth
    $REF/soap3-dp-builder $name.fa > /dev/null
    $REF/BGS-Build $name.fa.index > /dev/null
    for mm in 0 1 2 3 4; do $REF/ref_harness search $INI $name.fa.index $name.fq $L $mm search_mm$mm.txt > /dev/null 2>&1; done
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 4 -b 2 -o sam_s4 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 2 -b 2 -o sam_s2 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 2 -h 1 -b 2 -o sam_h1_s2 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 0 -h 1 -b 2 -o sam_h1_s0 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 3 -h 3 -b 2 -o sam_h3_s3 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 3 -h 4 -b 2 -p -o sam_h4_s3 > /dev/null 2>&1
    sha256sum $name.fa.index.lkt $name.fa.index.rev.lkt | awk '{print $1}' > lkt.sha256
    for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp $name.fa.index.$f $OUT/$name/index.$f; done
    cp $name.fa $OUT/$name/genome.fa; cp $name.fq $OUT/$name/reads.fq; cp $name.truth $OUT/$name/reads.truth; cp lkt.sha256 $OUT/$name/
    for mm in 0 1 2 3 4; do gzip -9n < search_mm$mm.txt > $OUT/$name/search_mm$mm.txt.gz; done
    grep -v '^@PG' sam_s4.gout.1 | gzip -9n > $OUT/$name/sam_s4.sam.gz
    grep -v '^@PG' sam_s2.gout.1 | gzip -9n > $OUT/$name/sam_s2.sam.gz
    for v in h1_s2 h1_s0 h3_s3 h4_s3; do grep -v '^@PG' sam_$v.gout.1 | gzip -9n > $OUT/$name/sam_$v.sam.gz; done
  )
}

# long-read mode (-L > 120: the first 100 bases are searched, the rest verified): SAM only
mkl() {  # name genome_args read_args maxlen
  local name=$1; shift
  local gargs=$1; shift
  local rargs=$1; shift
  local L=$1; shift
  mkdir -p $OUT/$name
  ( cd $W && python $ROOT/tools/simulate.py $gargs --fasta $name.fa $rargs --fastq $name.fq --truth $name.truth
    $REF/soap3-dp-builder $name.fa > /dev/null
    $REF/BGS-Build $name.fa.index > /dev/null
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 3 -b 2 -o sam_s3 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 1 -b 2 -p -o sam_s1 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 2 -h 1 -b 2 -o sam_h1_s2 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 2 -h 3 -b 2 -o sam_h3_s2 > /dev/null 2>&1
    $REF/ref_harness sam $INI single $name.fa.index $name.fq -L $L -s 3 -h 4 -b 2 -o sam_h4_s3 > /dev/null 2>&1
    sha256sum $name.fa.index.lkt $name.fa.index.rev.lkt | awk '{print $1}' > lkt.sha256
    for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp $name.fa.index.$f $OUT/$name/index.$f; done
    cp $name.fa $OUT/$name/genome.fa; cp $name.fq $OUT/$name/reads.fq; cp $name.truth $OUT/$name/reads.truth; cp lkt.sha256 $OUT/$name/
    for v in s3 s1 h1_s2 h3_s2 h4_s3; do grep -v '^@PG' sam_$v.gout.1 | gzip -9n > $OUT/$name/sam_$v.sam.gz; done
  )
}

# paired-end, DP stages off (pair ... -s k): SAM only
mkp() {  # name genome_args pair_args maxlen
  local name=$1; shift
  local gargs=$1; shift
  local rargs=$1; shift
  local L=$1; shift
  mkdir -p $OUT/$name
  ( cd $W && python $ROOT/tools/simulate.py $gargs --fasta $name.fa $rargs --fastq ${name}_1.fq --fastq2 ${name}_2.fq --truth $name.truth
    $REF/soap3-dp-builder $name.fa > /dev/null
    $REF/BGS-Build $name.fa.index > /dev/null
    P="$REF/ref_harness sam $INI pair $name.fa.index ${name}_1.fq ${name}_2.fq -L $L -u 500 -v 200 -b 2"
    $P -s 3 -o pe_s3 > /dev/null 2>&1
    $P -s 1 -p -o pe_s1 > /dev/null 2>&1
    $P -s 4 -o pe_s4 > /dev/null 2>&1
    $P -s 2 -h 1 -o pe_h1_s2 > /dev/null 2>&1
    $P -s 3 -h 3 -o pe_h3_s3 > /dev/null 2>&1
    $P -s 2 -h 4 -o pe_h4_s2 > /dev/null 2>&1
    sha256sum $name.fa.index.lkt $name.fa.index.rev.lkt | awk '{print $1}' > lkt.sha256
    for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp $name.fa.index.$f $OUT/$name/index.$f; done
    cp $name.fa $OUT/$name/genome.fa; cp ${name}_1.fq $OUT/$name/reads_1.fq; cp ${name}_2.fq $OUT/$name/reads_2.fq; cp $name.truth $OUT/$name/reads.truth
    cp lkt.sha256 $OUT/$name/
    for v in s3 s1 s4 h1_s2 h3_s3 h4_s2; do grep -v '^@PG' pe_$v.gout.1 | gzip -9n > $OUT/$name/pe_$v.sam.gz; done
  )
}

if [ -z "${S3_FIXTURE_ONLY:-}" ]; then     # S3_FIXTURE_ONLY=dp: only the DP-stage fixtures below
mk g1 "--genome-len 200000 --chroms 3 --n-runs 4 --repeats 6 --genome-seed 1" "--reads 2000 --read-len 100 --read-seed 2026" 100
mk g2 "--genome-len 60000 --chroms 2 --n-runs 2 --repeats 40 --genome-seed 7" "--reads 1500 --read-len 100 --var-len 25 120 --read-seed 11" 120
mkl g3 "--genome-len 120000 --chroms 2 --n-runs 4 --repeats 12 --genome-seed 5" "--reads 1200 --var-len 121 200 --read-seed 31 --mix 0.3 0.2 0.15 0.1 0.08 0.07 0.05 0.03 0.02" 200
mkp g4 "--genome-len 200000 --chroms 3 --n-runs 4 --repeats 8 --genome-seed 9" "--pairs 1200 --read-len 100 --read-seed 77" 100
# routing of unpaired pairs with the DP stages on (hostKernel's needProceedDP branch): which end goes to which DP stage, in which order
( cd $W && $REF/ref_harness sam $INI pair g4.fa.index $OUT/g4/reads_1.fq $OUT/g4/reads_2.fq -u 500 -v 100 -L 100 -o g4dp > /dev/null 2>&1
  gzip -9n < g4dp.routing > $OUT/g4/pe_routing.txt.gz
  mkdir -p $OUT/g5
  python $ROOT/tools/simulate_routing_case.py --fasta g5.fa --fastq1 g5_1.fq --fastq2 g5_2.fq
  $REF/soap3-dp-builder g5.fa > /dev/null; $REF/BGS-Build g5.fa.index > /dev/null
  $REF/ref_harness sam $INI pair g5.fa.index g5_1.fq g5_2.fq -u 500 -v 100 -L 100 -o g5dp > /dev/null 2>&1
  for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp g5.fa.index.$f $OUT/g5/index.$f; done
  cp g5_1.fq $OUT/g5/reads_1.fq; cp g5_2.fq $OUT/g5/reads_2.fq; gzip -9n < g5dp.routing > $OUT/g5/pe_routing.txt.gz
)
# routing of unpaired pairs with the DP stages on (hostKernel's needProceedDP branch): which end goes to which DP stage, in which order
( cd $W && $REF/ref_harness sam $INI pair g4.fa.index g4_1.fq g4_2.fq -u 500 -v 100 -L 100 -o g4dp > /dev/null 2>&1
  gzip -9n < g4dp.routing > $OUT/g4/pe_routing.txt.gz
  mkdir -p $OUT/g5
  python $ROOT/tools/simulate_routing_case.py --fasta g5.fa --fastq1 g5_1.fq --fastq2 g5_2.fq
  $REF/soap3-dp-builder g5.fa > /dev/null; $REF/BGS-Build g5.fa.index > /dev/null
  $REF/ref_harness sam $INI pair g5.fa.index g5_1.fq g5_2.fq -u 500 -v 100 -L 100 -o g5dp > /dev/null 2>&1
  for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp g5.fa.index.$f $OUT/g5/index.$f; done
  cp g5_1.fq $OUT/g5/reads_1.fq; cp g5_2.fq $OUT/g5/reads_2.fq; gzip -9n < g5dp.routing > $OUT/g5/pe_routing.txt.gz
)
# g6: the ends of the text -- reads that hug / overhang them with runs of A (padded lookup keys), a repeat, a sequence boundary
mkdir -p $OUT/g6
( cd $W && python $ROOT/tools/simulate_edge_case.py --fasta g6.fa --fastq g6.fq
  $REF/soap3-dp-builder g6.fa > /dev/null; $REF/BGS-Build g6.fa.index > /dev/null
  for mm in 0 1 2 3 4; do $REF/ref_harness search $INI g6.fa.index g6.fq 100 $mm search_mm$mm.txt > /dev/null 2>&1; gzip -9n < search_mm$mm.txt > $OUT/g6/search_mm$mm.txt.gz; done
  $REF/ref_harness sam $INI single g6.fa.index g6.fq -L 100 -s 4 -b 2 -o sam_s4 > /dev/null 2>&1
  $REF/ref_harness sam $INI single g6.fa.index g6.fq -L 100 -s 2 -h 1 -p -b 2 -o sam_h1_s2 > /dev/null 2>&1
  for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp g6.fa.index.$f $OUT/g6/index.$f; done
  cp g6.fq $OUT/g6/reads.fq
  grep -v '^@PG' sam_s4.gout.1 | gzip -9n > $OUT/g6/sam_s4.sam.gz; grep -v '^@PG' sam_h1_s2.gout.1 | gzip -9n > $OUT/g6/sam_h1_s2p.sam.gz
)
# output stage of the single-read DP: reads + synthetic DP result vectors -> the reference's own SAM records
mkdir -p $OUT/dp1
( cd $W && python $ROOT/tools/simulate.py --genome-len 90000 --chroms 3 --n-runs 4 --genome-seed 21 --fasta dp1.fa
  $REF/soap3-dp-builder dp1.fa > /dev/null
  $REF/BGS-Build dp1.fa.index > /dev/null
  python $ROOT/tools/simulate_dp_results.py --index dp1.fa.index --fastq dp1.fq --results dp1.results
  P="$REF/ref_harness dpsingle $INI dp1.results single dp1.fa.index dp1.fq -L 100"
  $P -o dp_h2 > /dev/null 2>&1; $P -h 1 -o dp_h1 > /dev/null 2>&1; $P -p -o dp_h2_p > /dev/null 2>&1; $P -h 1 -p -o dp_h1_p > /dev/null 2>&1
  for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp dp1.fa.index.$f $OUT/dp1/index.$f; done
  cp dp1.fq $OUT/dp1/reads.fq; cp dp1.results $OUT/dp1/results.txt
  for v in h2 h1 h2_p h1_p; do grep -v '^@PG' dp_$v.dpout.1 | gzip -9n > $OUT/dp1/dpout_$v.sam.gz; done
)
# output stage of the paired DP stages (default DP, deep DP): synthetic result vectors on the dp1 index -> the reference's own records
mkdir -p $OUT/dp2
( cd $W && python $ROOT/tools/simulate_dp_pair_results.py --index dp1.fa.index --fastq1 dp2_1.fq --fastq2 dp2_2.fq --results dp2.results --stats dp2.stats
  sed 's/^BWALikeScore *= *1/BWALikeScore=0/' $INI > nobwa.ini
  run() { $REF/ref_harness dppair $1 dp2.results dp2.stats pair dp1.fa.index dp2_1.fq dp2_2.fq -u 500 -v 100 -L 100 $3 -o r_$2 > /dev/null 2>&1; }
  run $INI h2 ""; run $INI h1 "-h 1"; run $INI h2p "-p"; run $INI h1p "-h 1 -p"; run $INI h3 "-h 3"; run $INI h4p "-h 4 -p"; run nobwa.ini nb2 "-p"; run nobwa.ini nb1 "-h 1"
  cp dp2_1.fq $OUT/dp2/reads_1.fq; cp dp2_2.fq $OUT/dp2/reads_2.fq; cp dp2.results $OUT/dp2/results.txt; cp dp2.stats $OUT/dp2/stats.txt
  for v in h2 h1 h2p h1p h3 h4p nb2 nb1; do grep -v '^@PG' r_$v.dpout.1 | gzip -9n > $OUT/dp2/dpout_$v.sam.gz; done
)
fi  # S3_FIXTURE_ONLY

# ---------------------------------------------------------------------------------------------------------------------------------
# DP stages ON: the reference's own DP engines (host halves of DV-DPfunctions.cu and callers, compiled by oracle/ref.mk) run behind the
# BWT ladder.  Per run: the three SAM files (.gout.1 / .dpout.1 / .unpair) and every DP problem the engines submitted (packer output +
# what the decode stage received), sorted.
dprun() {  # outdir tag index mode(single|pair) reads... -- options
  local dir=$1 tag=$2; shift 2
  ( cd $W && rm -f $tag.gout.1 $tag.dpout.1 $tag.unpair $tag.problems
    REFDP_DUMP=$tag.problems $REF/ref_harness sam $INI "$@" -o $tag > $tag.log 2>&1
    for k in gout.1 dpout.1 unpair; do if [ -f $tag.$k ]; then grep -v '^@PG' $tag.$k | gzip -9n > $dir/dp_$tag.${k%.1}.sam.gz; fi; done
    grep '^P' $tag.problems | LC_ALL=C sort | gzip -9n > $dir/dp_$tag.problems.txt.gz )
}
( cd $W
  [ -f g4.fa.index.bwt ] || { cp $OUT/g4/genome.fa g4.fa; $REF/soap3-dp-builder g4.fa > /dev/null; $REF/BGS-Build g4.fa.index > /dev/null; }
  [ -f g5.fa.index.bwt ] || { python $ROOT/tools/simulate_routing_case.py --fasta g5.fa --fastq1 g5_1.fq --fastq2 g5_2.fq; $REF/soap3-dp-builder g5.fa > /dev/null; $REF/BGS-Build g5.fa.index > /dev/null; }
  cmp g5_1.fq $OUT/g5/reads_1.fq
  # g7: config C3's shape -- 2 x 150 bp, -u 500 -v 1; 10 % of the pairs carry an indel in one mate, 5 % are damaged in both
  mkdir -p $OUT/g7
  python $ROOT/tools/simulate.py --genome-len 300000 --chroms 3 --n-runs 4 --repeats 10 --genome-seed 13 --fasta g7.fa --pairs 1500 --read-len 150 --read-seed 91 \
         --fastq g7_1.fq --fastq2 g7_2.fq --truth g7.truth --indel-frac 0.10 --damaged-frac 0.05
  $REF/soap3-dp-builder g7.fa > /dev/null; $REF/BGS-Build g7.fa.index > /dev/null
  for f in bwt rev.bwt fmv.gpu rev.fmv.gpu sa pac ann amb tra; do cp g7.fa.index.$f $OUT/g7/index.$f; done
  cp g7.fa $OUT/g7/genome.fa; cp g7_1.fq $OUT/g7/reads_1.fq; cp g7_2.fq $OUT/g7/reads_2.fq; cp g7.truth $OUT/g7/reads.truth
  sha256sum g7.fa.index.lkt g7.fa.index.rev.lkt | awk '{print $1}' > $OUT/g7/lkt.sha256
  # 2 x 200 bp pairs on the same genome: above 150 bases the BWT stage is skipped and every pair goes to the deep DP (SOAP3-DP.cu:658-662)
  python $ROOT/tools/simulate.py --genome-len 300000 --chroms 3 --n-runs 4 --repeats 10 --genome-seed 13 --fasta g7x.fa --pairs 300 --read-len 200 --read-seed 5 \
         --fastq g7_long_1.fq --fastq2 g7_long_2.fq --insert-mean 450 --indel-frac 0.15 --damaged-frac 0.05
  cmp g7.fa g7x.fa; cp g7_long_1.fq $OUT/g7/long_1.fq; cp g7_long_2.fq $OUT/g7/long_2.fq
)
# g7 with the DP stages off, every report type (the paired ladder at 150 bp, wordPerQuery 16)
( cd $W
  P="$REF/ref_harness sam $INI pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2"
  $P -s 3 -o pe_s3 > /dev/null 2>&1; $P -s 3 -h 1 -o pe_h1_s3 > /dev/null 2>&1; $P -s 3 -h 3 -o pe_h3_s3 > /dev/null 2>&1; $P -s 3 -h 4 -p -o pe_h4_s3 > /dev/null 2>&1
  for v in s3 h1_s3 h3_s3 h4_s3; do grep -v '^@PG' pe_$v.gout.1 | gzip -9n > $OUT/g7/pe_$v.sam.gz; done
)
dprun $OUT/g7 h2  pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2
dprun $OUT/g7 h2p pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2 -p
# the same run as the reference decides it on a SMALL device (decideConfiguration's scheme 2, DV-DPfunctions.cu:600-612: score pass over the whole window,
# table of L + L/2 + 8 columns): the default DP's 650-column windows take it (>= 3 x read length), the deep DP's 224-column windows do not
REFDP_SCHEME=2 dprun $OUT/g7 h2s2 pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2
dprun $OUT/g7 h1  pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2 -h 1
dprun $OUT/g7 h3  pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2 -h 3
dprun $OUT/g7 h4  pair g7.fa.index g7_1.fq g7_2.fq -L 150 -u 500 -v 1 -b 2 -h 4
dprun $OUT/g7 long pair g7.fa.index g7_long_1.fq g7_long_2.fq -L 200 -u 600 -v 100 -b 2
dprun $OUT/g7 se   single g7.fa.index g7_1.fq -L 150 -b 2
dprun $OUT/g7 se_h1 single g7.fa.index g7_2.fq -L 150 -b 2 -h 1 -p
rm -f $OUT/g7/dp_h2p.problems.txt.gz $OUT/g7/dp_h1.problems.txt.gz $OUT/g7/dp_h4.problems.txt.gz      # same problems as h2 / h3
dprun $OUT/g4 h2  pair g4.fa.index $OUT/g4/reads_1.fq $OUT/g4/reads_2.fq -L 100 -u 500 -v 100 -b 2
dprun $OUT/g4 h1  pair g4.fa.index $OUT/g4/reads_1.fq $OUT/g4/reads_2.fq -L 100 -u 500 -v 100 -b 2 -h 1
rm -f $OUT/g4/dp_h1.problems.txt.gz
dprun $OUT/g5 h2  pair g5.fa.index g5_1.fq g5_2.fq -L 100 -u 500 -v 100 -b 2
# the g1 index with a SAMPLED suffix array (SaValueFreq=4, README.md:55-65): only the .sa file differs
( cd $W && mkdir -p sa4 && cd sa4 && cp $OUT/g1/genome.fa g1.fa && cp $REF/soap3-dp-builder . && sed 's/^SaValueFreq=1;/SaValueFreq=4;/' $ROOT/oracle/builder.ini > soap3-dp-builder.ini
  ./soap3-dp-builder g1.fa > /dev/null && cmp g1.fa.index.bwt $OUT/g1/index.bwt && cp g1.fa.index.sa $OUT/g1/index_sa4.sa )
( cd $W; [ -f g1.fa.index.bwt ] || { cp $OUT/g1/genome.fa g1.fa; $REF/soap3-dp-builder g1.fa > /dev/null; $REF/BGS-Build g1.fa.index > /dev/null; } )
dprun $OUT/g1 se  single g1.fa.index $OUT/g1/reads.fq -L 100 -b 2
if [ -z "${S3_FIXTURE_ONLY:-}" ]; then
( cd $W && $REF/ref_harness models models.txt && gzip -9n < models.txt > $OUT/models.txt.gz )
( cd $W && $REF/ref_harness dpparams $INI dpparams.txt && gzip -9n < dpparams.txt > $OUT/dpparams.txt.gz )
fi
# BAM files for the `-bam` input tests, written by the reference's samtools library (oracle/_ref/ref_sam2bam)
if [ -z "${S3_FIXTURE_OUT:-}" ]; then python $ROOT/tests/golden/make_bam_fixtures.py; fi
du -sh $OUT
