#!/bin/bash
# debugging aid: after debug_every_route.sh -- the pairs whose dpout records differ, alone, with DP problem dumps of both sides compared as sets
cd /tmp/every_route || exit 1
R=$GRAFT_REPO_ROOT
python3 - <<'PY'
import re, collections, itertools
def load(fn):
    d=collections.defaultdict(list)
    for l in open(fn):
        if l.startswith('@'): continue
        l=re.sub(r"\tRG:Z:[^\t\n]*","",l.rstrip("\n"))
        f=l.split("\t")
        d[(f[0], int(f[1])&0xC0)].append(l)
    return d
a,b=load('ref.dpout.1'),load('mine.dpout.1')
names=sorted(set(k[0] for k in set(a)|set(b) if sorted(a.get(k,[]))!=sorted(b.get(k,[]))))
print(len(names), names)
for fn,out in (('r_1.fq','t_1.fq'),('r_2.fq','t_2.fq')):
    with open(fn) as f, open(out,'w') as o:
        while True:
            rec=list(itertools.islice(f,4))
            if not rec: break
            if rec[0][1:].split()[0].split('/')[0] in names: o.writelines(rec)
PY
REFDP_DUMP=$PWD/tref_p.txt $R/oracle/_ref/ref_harness sam $R/soap3-dp_amd/host/soap3-dp.ini pair g.fa.index t_1.fq t_2.fq -L 150 -u 500 -v 1 -b 2 -o tref > tref.log 2>&1; grep harness tref.log
S3DP_DUMP_DP=$PWD/tmine_p.txt S3DP_VERBOSE=1 $R/soap3-dp_amd/soap3-dp pair g.fa.index t_1.fq t_2.fq -L 150 -u 500 -v 1 -b 2 -o tmine > tmine.log 2>&1; grep "Number of pairs\|deep DP stage" tmine.log | cut -c1-250
python3 - <<'PY'
import collections
def probs(fn):
    sec=None; d=collections.Counter(); per=collections.Counter()
    for l in open(fn):
        if l.startswith('#'): sec=l.strip(); continue
        if l.startswith('P'): d[l.strip()]+=1; per[sec]+=1
    return d, per
r,rp=probs('tref_p.txt'); m,mp=probs('tmine_p.txt')
print("ref per stage",dict(rp)); print("mine per stage",dict(mp))
onlyr=[k for k in r if k not in m]; onlym=[k for k in m if k not in r]
print("distinct problems ref",len(r),"mine",len(m),"only ref",len(onlyr),"only mine",len(onlym))
for k in onlyr[:3]: print("ONLY REF ",k[:330])
for k in onlym[:3]: print("ONLY MINE",k[:330])
PY
mkdir -p $R/gpurun_out/r05dbg4; cp t_1.fq t_2.fq tref.log tmine.log $R/gpurun_out/r05dbg4/; gzip -c tref_p.txt > $R/gpurun_out/r05dbg4/tref_p.txt.gz; gzip -c tmine_p.txt > $R/gpurun_out/r05dbg4/tmine_p.txt.gz; cp tref.dpout.1 tmine.dpout.1 $R/gpurun_out/r05dbg4/
