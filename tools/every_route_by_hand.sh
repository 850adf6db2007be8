#!/bin/bash
# debugging aid: the every-route scale case (tests/test_gpu_cli_vs_oracle_scale.py::...[pair_every_route]) by hand, with the programs' own count lines
ulimit -c 0
W=/tmp/every_route; rm -rf $W; mkdir -p $W; cd $W
R=$GRAFT_REPO_ROOT
python3 $R/tools/simulate.py --genome-len 3000000 --chroms 3 --n-runs 4 --repeats 16 --genome-seed 55 --fasta g.fa --family-copies 600 --family-len 700 --family-div 0.004 --pairs 30000 --read-len 150 --read-seed 5 --fastq r_1.fq --fastq2 r_2.fq --indel-frac 0.10 --damaged-frac 0.05
cp $R/oracle/_ref/soap3-dp-builder.ini . ; $R/oracle/_ref/soap3-dp-builder g.fa > /dev/null; $R/oracle/_ref/BGS-Build g.fa.index > /dev/null
$R/oracle/_ref/ref_harness sam $R/soap3-dp_amd/host/soap3-dp.ini pair g.fa.index r_1.fq r_2.fq -L 150 -u 500 -v 1 -b 2 -o ref > ref.log 2>&1; grep harness ref.log
for cfg in "A=1" "$@"; do
  echo "== $cfg"
  env $cfg S3DP_CHUNK_READS=16384 ${BIN:-$R/soap3-dp_amd/soap3-dp} pair g.fa.index r_1.fq r_2.fq -L 150 -u 500 -v 1 -b 2 -o mine > mine.log 2>&1; grep "Number of pairs\|Pairs left\|Overall" mine.log
  for k in gout.1 dpout.1 unpair; do echo "$k ref $(grep -vc '^@' ref.$k) mine $(grep -vc '^@' mine.$k)"; done
done
grep -v "^@" ref.dpout.1 | cut -f1 | sort -u > a.txt; grep -v "^@" mine.dpout.1 | cut -f1 | sort -u > b.txt; comm -23 a.txt b.txt | head -5 > only_ref.txt; cat only_ref.txt
n=$(head -1 only_ref.txt); grep -P "^$n\t" ref.dpout.1 | cut -f1-9; grep -P "^$n\t" mine.unpair | cut -f1-9; grep -P "^$n\t" ref.routing 2>/dev/null | head -3
grep -c "^N" ref.routing; grep "^N" ref.routing | cut -d' ' -f2 | sort -u | wc -l
