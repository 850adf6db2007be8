#!/bin/bash
# round 4: does the run-to-run spread of the paired command line follow where its threads and host buffers land (NUMA)?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04u; mkdir -p $O /tmp/s3dp_ix
{ lscpu | grep -i "numa\|socket\|model name\|thread"; for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done; which taskset numactl; cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list"; rocm-smi --showtoponuma 2>&1 | tail -8; } > $O/topo.txt 2>&1
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3_10m.json
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -b 2"
export S3DP_VERBOSE=1
run() { local n=$1; shift; timeout 120 "$@" $CLI -o /dev/shm/c3o_$n > $O/$n.log 2>&1; echo "$n rc=$? $(grep 'Alignment + output' $O/$n.log | sed 's/.*: *//') | $(grep 'last chunk searched\|DP stage threads done' $O/$n.log | sed 's/.*member 0: //; s/ after the index was resident//' | tr -s ' ' | tr '\n' ';')"; rm -f /dev/shm/c3o_$n*; }
GN=$(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | sort -u | grep -v "^-1" | head -1)
echo "gpu numa node: $GN"
NODES=$(ls -d /sys/devices/system/node/node* | wc -l)
for k in 1 2 3; do run free_$k; done
for nd in $(seq 0 $((NODES-1))); do
  cl=$(cat /sys/devices/system/node/node$nd/cpulist)
  for k in 1 2; do run node${nd}_$k taskset -c $cl; done
done
