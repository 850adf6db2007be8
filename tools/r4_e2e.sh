#!/bin/bash
# round 4: the command line end to end (FASTQ in, SAM out) on the round-3 workloads, and the other repeat models through bench.py's robustness leg
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
S3DP_VERBOSE=1 timeout 600 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 > $O/e2e_se_uniform.json 2> $O/e2e_se_uniform.log; echo "uniform rc=$?"; cut -c1-900 $O/e2e_se_uniform.json; echo
S3DP_VERBOSE=1 timeout 600 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; echo "repeat rc=$?"; cut -c1-900 $O/e2e_se_repeat.json; echo
timeout 600 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 --default-mode 1 > $O/e2e_se_default_mode.json 2> /dev/null; echo "default rc=$?"; cut -c1-700 $O/e2e_se_default_mode.json; echo
timeout 900 python3 bench.py --steps 3 --warmup 1 --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --exact-rank 0 --cpu-reference 0 --check 2048 --robustness old,satellite-heavy > $O/bench_robustness.json 2> $O/bench_robustness.log; echo "robust rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04k/bench_robustness.json"))
for k, v in d["extras"]["repeat_models"].items():
    print(k, {x: v.get(x) for x in ("M_reads_per_s", "ms_per_step", "kernel_ms_per_step", "rank_queries_per_read", "occurrences", "error")}, v.get("truth_check", {}).get("ok"), v.get("oracle_check", {}).get("bit_exact"))
PY
