#!/bin/bash
# round 4: `single -L 150 -h 1` on 10 M reads with the smaller default chunk of that mode (ran out of device memory at 1 Mi reads per chunk)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h2; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 400 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
timeout 150 soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -h 1 -b 2 -o /tmp/c3w/o_h1 > $O/h1.log 2>&1; echo "h1 rc=$? $(grep 'Alignment + output' $O/h1.log | sed 's/.*: *//') $(grep 'Device time\|out of memory\|2^32\|do not fit' $O/h1.log | cut -c1-200)"
