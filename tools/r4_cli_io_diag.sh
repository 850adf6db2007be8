#!/bin/bash
# round 4: where the paired command line's time goes at the end of a run (S3DP_VERBOSE timeline), output on /tmp against /dev/shm, new files against files written over
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04t; mkdir -p $O /tmp/s3dp_ix
{ free -g; df -h /tmp /dev/shm; mount | grep -E " /tmp | / " ; for f in dirty_ratio dirty_background_ratio dirty_bytes dirty_background_bytes dirty_expire_centisecs; do echo "$f=$(cat /proc/sys/vm/$f)"; done; nproc; } > $O/box.txt 2>&1
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3_10m.json
grep Dirty /proc/meminfo
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
export S3DP_VERBOSE=1
run() { timeout 120 $CLI -o $2 > $O/$1.log 2>&1; echo "$1 rc=$? $(grep 'Alignment + output' $O/$1.log) dirty=$(grep Dirty: /proc/meminfo | tr -s ' ' | cut -d' ' -f2) kB"; grep "last chunk searched\|DP stage threads done\|first output file closed\|chunk loop done" $O/$1.log | sed 's/.*member 0: //'; }
run tmp_new_1 /tmp/c3w/o1
run tmp_new_2 /tmp/c3w/o2
run tmp_over_1 /tmp/c3w/o1
sync; echo "synced"
run tmp_new_3_after_sync /tmp/c3w/o3
rm -f /tmp/c3w/o1* /tmp/c3w/o2*
mkdir -p /dev/shm/c3o
run shm_new_1 /dev/shm/c3o/o1
run shm_over_1 /dev/shm/c3o/o1
rm -rf /dev/shm/c3o
run devnull /dev/null
