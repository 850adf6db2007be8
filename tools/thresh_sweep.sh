cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03t /tmp/ix
Q="--index-dir /tmp/ix --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 --steps 2 --warmup 1"
timeout 600 python3 bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2>&1
for cfg in "1:4096" "1:2048" "1:1024" "1:512" "512:1024" "256:1024" "256:512" "2048:4096"; do
  h=${cfg%%:*}; r=${cfg##*:}
  S3DP_HEAVY=$h S3DP_HEAVY_RANK=$r timeout 300 python3 bench.py $Q 2> gpurun_out/r03t/log_$h_$r.txt | python3 -c "
import sys, json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('heavy=$h rank=$r ms_per_step %.1f stepping %.1f ce %.1f ok %s' % (d['ms_per_step'], r['stepping_kernel_ms_per_step'], r['check_extend_kernel_ms_per_step'], d['truth_check']['ok']))
"
done
