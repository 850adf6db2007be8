#!/bin/bash
# round 4: adoption across waves -- parity subset, resident rate at 1 Mi / 4 Mi / 10 M reads, streamed figures (every step under its own timeout)
cd "$GRAFT_REPO_ROOT"
S3DP_SMALL_SIZES="1048576 4194304" timeout 400 tools/r4_small_batches.sh 2>&1 | grep "^reads\|unit pass" | cut -c1-300
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
timeout 500 python3 bench.py --index-dir /tmp/ix --steps 5 --warmup 2 --uniform 0 --extras 0 --cpu-budget 0 --exact-rank 0 --robustness= --cpu-reference 0 2> /tmp/b.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['oracle_check']['bit_exact'], d['truth_check']['ok'], d['streamed']['vs_resident'], d['streamed']['batches_of_1Mi']['vs_resident'])"; grep "resident\|streamed" /tmp/b.log
