#!/bin/bash
# Measurement set (rounds 2 and 3) for the GPU box (run through gpurun from the repository root; writes small summaries under gpurun_out/r02_final/, which
# are then copied into profiles/).  Every step has its own timeout; the benchmark index is built once and kept as files under /tmp.
#   tools/run_profiles.sh [bench|stats|pmc|dp|e2e|pairs ...]     (default: all)
set -u
OUT=${S3DP_PROFILE_OUT:-gpurun_out/r04}
IX=/tmp/s3dp_ix
mkdir -p $OUT $IX
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
STEPS="${*:-bench stats pmc dp e2e pairs}"
QUICK="--index-dir $IX --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --exact-rank 0 --robustness= --cpu-reference 0 --repeat-frac ${S3DP_PROFILE_REPEAT_FRAC:-0.5}"
want() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }

timeout 900 python3 bench.py --index-dir $IX --repeat-frac ${S3DP_PROFILE_REPEAT_FRAC:-0.5} --build-only 1 > /dev/null 2> $OUT/index_build.log; echo "index build rc=$?"

if want bench; then   # the default run, as the driver runs it
  timeout 1500 python3 bench.py --index-dir $IX --steps 5 > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc=$?"; grep "resident:\|streamed:" $OUT/bench.log
fi
if want stats; then   # kernel stats of the timed passes (rocprofv3 --kernel-trace --stats)
  rm -rf /tmp/prof_stats
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 bench.py $QUICK --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log
  echo "stats rc=$?"; f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_bench.csv && grep "s3::" $OUT/kernel_stats_bench.csv | head -16 | cut -c1-160
fi
if want pmc; then     # HBM traffic of the search kernels: FETCH_SIZE in its own pass, one ladder (4 phases), no warm-up
  rm -rf /tmp/prof_pmc
  timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_pmc -o pmc -- python3 bench.py $QUICK --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.log
  echo "pmc rc=$?"; f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/pmc_fetch_${S3DP_PROFILE_REPEAT_FRAC:-0.5}.csv
  if [ -n "$f" ]; then python3 tools/pmc_summary.py "$f" firstKernel > $OUT/pmc_fetch_first.txt; python3 tools/pmc_summary.py "$f" unitKernel > $OUT/pmc_fetch_unit.txt; python3 tools/pmc_summary.py "$f" ceKernel > $OUT/pmc_fetch_ce.txt; tail -4 $OUT/pmc_fetch_unit.txt; fi
  rm -rf /tmp/prof_pmc
  timeout 1200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/prof_pmc -o pmc -- python3 bench.py $QUICK --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.log
  echo "pmc sq rc=$?"; f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_summary.py "$f" firstKernel > $OUT/pmc_sq_first.txt; python3 tools/pmc_summary.py "$f" unitKernel > $OUT/pmc_sq_unit.txt; python3 tools/pmc_summary.py "$f" ceKernel > $OUT/pmc_sq_ce.txt; tail -4 $OUT/pmc_sq_unit.txt; fi
fi
if want dp; then      # DP kernel: GCUPS, kernel stats, counters on 4096 problems per shape
  timeout 600 python3 tools/bench_dp.py > $OUT/dp_gcups.jsonl 2> $OUT/dp_gcups.log; echo "dp rc=$?"; cat $OUT/dp_gcups.jsonl | cut -c1-220
  rm -rf /tmp/prof_dp
  S3DP_BENCH_DP_PROBLEMS=4096 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/prof_dp -o dp -- python3 tools/bench_dp.py > /dev/null 2> $OUT/dp_pmc_sq.log
  f=$(find /tmp/prof_dp -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py "$f" dpFill > $OUT/dp_pmc_sq.txt && tail -3 $OUT/dp_pmc_sq.txt
  rm -rf /tmp/prof_dp2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp2 -o dp -- python3 tools/bench_dp.py > /dev/null 2> $OUT/dp_stats.log
  f=$(find /tmp/prof_dp2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/dp_kernel_stats.csv && head -6 $OUT/dp_kernel_stats.csv | cut -c1-60,100-200
  # (WRITE_SIZE and FETCH_SIZE do not fit one pass -- 2 + 3 of the 4 TCC slots: rocprofv3 aborts; the table traffic is algorithmic: 2.125 B per cell)
fi
if want e2e; then     # the command line, FASTQ in -> SAM out, GRCh38 scale: uniform genome (round 1's workload), repeat-rich genome, two members on one GPU
  timeout 900 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 > $OUT/e2e_se_uniform.json 2> $OUT/e2e_se_uniform.log; echo "e2e uniform rc=$?"; cat $OUT/e2e_se_uniform.json | cut -c1-600
  timeout 1200 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 > $OUT/e2e_se_repeat.json 2> $OUT/e2e_se_repeat.log; echo "e2e repeat rc=$?"; cat $OUT/e2e_se_repeat.json | cut -c1-600
fi
if want pairs; then   # config C3's shape through all DP stages
  timeout 1200 python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 > $OUT/e2e_pairs_dp.json 2> $OUT/e2e_pairs_dp.log; echo "pairs rc=$?"; cat $OUT/e2e_pairs_dp.json | cut -c1-700
fi
du -sh $OUT
