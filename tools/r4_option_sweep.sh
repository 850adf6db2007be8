#!/bin/bash
# round 4: the command line on the 10 M C3 pairs / their first mates under report types and modes that no benchmark leg runs -- looking for a path whose time is out of line
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
run() { local n=$1; shift; sync; timeout 100 soap3-dp_amd/soap3-dp "$@" -b 2 -o /tmp/c3w/o_$n > $O/$n.log 2>&1; echo "$n rc=$? $(grep 'Alignment + output' $O/$n.log | sed 's/.*: *//') $(grep 'Device time' $O/$n.log | cut -c1-120)"; rm -f /tmp/c3w/o_$n*; }
P="/tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1"
S="/tmp/c3w/g.index /tmp/c3w/r1.fq -L 150"
run pair_h1 pair $P -h 1
run pair_h4 pair $P -h 4
run pair_h3 pair $P -h 3
run pair_s2 pair $P -s 2
run pair_s4_h1 pair $P -s 4 -h 1
run single_h1 single $S -h 1
run single_h4 single $S -h 4
run single_h3_s3 single $S -h 3 -s 3
run single_p single $S -p
