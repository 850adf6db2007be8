mkdir -p gpurun_out/r03b /tmp/ix
timeout 600 python bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2>&1
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --index-dir /tmp/ix --steps 2 --warmup 1 --uniform 0 --extras 0 --streaming 0 --cpu-budget 0 --check 0 > gpurun_out/r03b/$name.json 2> gpurun_out/r03b/$name.log
  echo "$name: $(grep -o 'resident: [0-9.]* M.*' gpurun_out/r03b/$name.log | cut -c1-150)"; }
run b1024 S3DP_VERBOSE=1
run b512 S3DP_UNIT_BUDGET=512
run b2048 S3DP_UNIT_BUDGET=2048
grep "s3dp\] search" gpurun_out/r03b/b1024.log | tail -36 | cut -c1-100
