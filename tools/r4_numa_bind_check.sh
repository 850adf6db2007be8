#!/bin/bash
# round 4: the GPU suite, then the paired and single-end command lines with and without the binding to the device's NUMA node (fresh output files, write-back flushed between runs)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04v; mkdir -p $O /tmp/s3dp_ix
timeout 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
SE="soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -b 2"
export S3DP_VERBOSE=1
run() { local n=$1; shift; sync; timeout 120 "$@" -o /tmp/c3w/o_$n > $O/$n.log 2>&1; echo "$n rc=$? $(grep 'Alignment + output' $O/$n.log | sed 's/.*: *//') | $(grep 'host threads on\|last chunk searched\|DP stage threads done' $O/$n.log | sed 's/.*member 0: //; s/ after the index was resident//' | tr -s ' ' | tr '\n' ';')"; rm -f /tmp/c3w/o_$n*; }
for k in 1 2 3; do run bind_$k $CLI; done
export S3DP_NUMA_BIND=0
for k in 1 2; do run free_$k $CLI; done
unset S3DP_NUMA_BIND
for k in 1 2; do run se_bind_$k $SE; done
export S3DP_NUMA_BIND=0
for k in 1 2; do run se_free_$k $SE; done
