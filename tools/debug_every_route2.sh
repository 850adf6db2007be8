#!/bin/bash
# debugging aid: the pairs the deep DP of the reference aligns and the product's does not, alone, with both sides' DP problem dumps
ulimit -c 0
W=/tmp/every_route; cd $W || exit 1
R=$GRAFT_REPO_ROOT
grep -v "^@" ref.dpout.1 | cut -f1 | sort -u > a.txt; grep -v "^@" mine.dpout.1 | cut -f1 | sort -u > b.txt; comm -23 a.txt b.txt > only_ref.txt; wc -l only_ref.txt
python3 - <<'PY'
names=set(l.strip() for l in open('only_ref.txt'))
import itertools
for fn,out in (('r_1.fq','s_1.fq'),('r_2.fq','s_2.fq')):
    with open(fn) as f, open(out,'w') as o:
        while True:
            rec=list(itertools.islice(f,4))
            if not rec: break
            nm=rec[0][1:].split()[0].split('/')[0]
            if nm in names: o.writelines(rec)
PY
grep -c "^@p" s_1.fq
REFDP_DUMP=$W/ref_problems.txt REFDP_SEED_DUMP=$W/ref_seeds.txt $R/oracle/_ref/ref_harness sam $R/soap3-dp_amd/host/soap3-dp.ini pair g.fa.index s_1.fq s_2.fq -L 150 -u 500 -v 1 -b 2 -o sref > sref.log 2>&1; grep harness sref.log
S3DP_DUMP_DP=$W/mine_problems.txt S3DP_VERBOSE=1 $R/soap3-dp_amd/soap3-dp pair g.fa.index s_1.fq s_2.fq -L 150 -u 500 -v 1 -b 2 -o smine > smine.log 2>&1; grep "Number of pairs\|Pairs left" smine.log
grep -c "^P" ref_problems.txt mine_problems.txt; cat ref_seeds.txt | head; grep -v "^P" mine_problems.txt | head -20; grep -v "^P" ref_problems.txt | head -20
head -c 600 ref_problems.txt
mkdir -p $R/gpurun_out/r05dbg; gzip -c ref_problems.txt > $R/gpurun_out/r05dbg/ref_problems.txt.gz; gzip -c mine_problems.txt > $R/gpurun_out/r05dbg/mine_problems.txt.gz; cp s_1.fq s_2.fq sref.log smine.log ref_seeds.txt $R/gpurun_out/r05dbg/ 2>/dev/null
