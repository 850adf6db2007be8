#!/bin/bash
# round 4: `single -L 150 -h 1` (all valid, seeds searched without a hit limit) ran out of device memory on the repeat-rich genome with 1 Mi reads per chunk: does a smaller chunk fit?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 400 python3 tools/bench_c3.py --pairs 1048576 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_1m.json 2> $O/c3_1m.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
run() { local n=$1; shift; timeout 100 soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -h 1 -b 2 -o /tmp/c3w/o_$n > $O/$n.log 2>&1; echo "$n rc=$? $(grep 'Alignment + output' $O/$n.log | sed 's/.*: *//') $(grep 'Device time\|out of memory\|2^32' $O/$n.log | cut -c1-160)"; rm -f /tmp/c3w/o_$n*; }
run chunk_1mi
S3DP_CHUNK_READS=262144 run chunk_256ki
S3DP_CHUNK_READS=65536 run chunk_64ki
S3DP_LT_EXTRA=2 run lookup_15
S3DP_DP_THREADS=1 S3DP_POST_THREAD=0 run two_batches
rocm-smi --showmeminfo vram 2>&1 | grep -i "total\|used" | head -3
