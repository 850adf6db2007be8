#!/bin/bash
# round 4: FETCH_SIZE and SQ counters of the search kernels of one bench ladder at the end of the round (each pass under its own timeout)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
Q="--index-dir /tmp/s3dp_ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0"
rm -rf /tmp/prof_pmc
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_pmc -o pmc -- python3 bench.py $Q --steps 1 --warmup 0 > /dev/null 2> $O/pmc_fetch.log; echo "pmc rc=$?"
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then cp "$f" /tmp/pmc_fetch_0.5.csv; for k in first unit ce; do python3 tools/pmc_summary.py "$f" ${k}Kernel > $O/pmc_fetch_$k.txt; done; tail -4 $O/pmc_fetch_unit.txt; python3 tools/make_pmc_traffic.py "${S3DP_COMMIT:-unknown}" 0.5=/tmp/pmc_fetch_0.5.csv > $O/pmc_traffic.json; head -c 500 $O/pmc_traffic.json; fi
rm -rf /tmp/prof_pmc
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/prof_pmc -o pmc -- python3 bench.py $Q --steps 1 --warmup 0 > /dev/null 2> $O/pmc_sq.log; echo "pmc sq rc=$?"
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then for k in first unit ce; do python3 tools/pmc_summary.py "$f" ${k}Kernel > $O/pmc_sq_$k.txt; done; tail -4 $O/pmc_sq_unit.txt; fi
