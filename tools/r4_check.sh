cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ix gpurun_out/r04m
S3DP_SMALL_SIZES="1048576 4194304" timeout 300 tools/r4_small_batches.sh 2>&1 | grep "^reads" | cut -c1-200
timeout 300 python3 bench.py --index-dir /tmp/ix --steps 5 --warmup 2 --uniform 0 --extras 0 --cpu-budget 0 --exact-rank 0 --robustness= --cpu-reference 0 2> /tmp/b.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['oracle_check']['bit_exact'], d['truth_check']['ok'], d['streamed']['vs_resident'], d['streamed']['batches_of_1Mi']['vs_resident'])"; grep "resident\|streamed" /tmp/b.log
timeout 600 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/ix > gpurun_out/r04m/c3.json 2> gpurun_out/r04m/c3.log; grep -o "\"end_to_end.*lookup_chars\": 16}" gpurun_out/r04m/c3.json | cut -c1-1200; grep -o "\"bwt_stage_resident\": {[^}]*" gpurun_out/r04m/c3.json | cut -c1-300; grep -o "\"oracle_check\": {.*bit_exact[^}]*}" gpurun_out/r04m/c3.json | cut -c1-600
