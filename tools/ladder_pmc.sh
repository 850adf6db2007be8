#!/bin/bash
# round 5: counters of the search kernels of one bench ladder, each set in its own rocprofv3 pass (no tracing domains beside --pmc); run through gpurun
#   tools/ladder_pmc.sh <outdir> [env settings, comma separated]
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r05p}; mkdir -p $O /tmp/s3dp_ix
envs=""; [ -n "$2" ] && envs=$(echo "$2" | tr ',' ' ')
[ -n "$envs" ] && export $envs
timeout 400 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0"
pass() {   # name, counters...
  local name=$1; shift
  rm -rf /tmp/prof_pmc
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_pmc -o pmc -- python3 bench.py $Q --steps 1 --warmup 0 > /dev/null 2> $O/pmc_$name.log; echo "pmc $name rc=$?"
  local f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    for k in first unit ce; do python3 tools/pmc_summary.py "$f" "s3::${k}Kernel" > $O/pmc_${name}_$k.txt; done
    tail -4 $O/pmc_${name}_unit.txt
    [ "$name" = fetch ] && cp "$f" /tmp/pmc_fetch_0.5.csv && python3 tools/make_pmc_traffic.py "${S3DP_COMMIT:-unknown}" 0.5=/tmp/pmc_fetch_0.5.csv > $O/pmc_traffic.json
  fi
}
pass lanes SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
