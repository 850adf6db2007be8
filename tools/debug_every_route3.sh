#!/bin/bash
# debugging aid: after tools/debug_every_route.sh -- the dpout records that differ, field by field
cd /tmp/every_route || exit 1
python3 - <<'PY'
import re, collections
def load(fn):
    d=collections.defaultdict(list)
    for l in open(fn):
        if l.startswith('@'): continue
        l=re.sub(r"\tRG:Z:[^\t\n]*","",l.rstrip("\n"))
        f=l.split("\t")
        d[(f[0], int(f[1])&0xC0)].append(l)
    return d
a,b=load('ref.dpout.1'),load('mine.dpout.1')
n=0
for k in sorted(set(a)|set(b)):
    if sorted(a.get(k,[]))!=sorted(b.get(k,[])):
        n+=1
        if n>6: continue
        for x,y in zip(sorted(a.get(k,[])), sorted(b.get(k,[]))):
            fx,fy=x.split("\t"),y.split("\t")
            print(k, "fields differing:", [(i,fx[i][:200],fy[i][:200]) for i in range(min(len(fx),len(fy))) if fx[i]!=fy[i]], len(fx), len(fy))
        if len(a.get(k,[]))!=len(b.get(k,[])): print(k,"record counts",len(a.get(k,[])),len(b.get(k,[])))
print("differing keys",n)
PY
