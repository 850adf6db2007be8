#!/bin/bash
# round 4: single reads with the DP stage on through the post thread -- the command-line parity tests, then 10 M 150-base and 100-base reads in the default mode
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O /tmp/s3dp_ix
timeout 900 python3 -m pytest tests/test_gpu_cli_sam.py tests/test_gpu_cli_vs_oracle_scale.py tests/test_gpu_dp_pipeline.py tests/test_gpu_group.py -m gpu -x -q > $O/pytest_cli.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_cli.log)"
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
cut -c1-100 /tmp/c3w/r1.fq | awk 'NR%4==2 || NR%4==0 {print substr($0,1,100); next} {print}' > /tmp/c3w/r100.fq
export S3DP_VERBOSE=1
run() { local n=$1; shift; sync; timeout 200 "$@" -o /tmp/c3w/o_$n > $O/$n.log 2>&1; echo "$n rc=$? $(grep 'Alignment + output\|Device time\|Chunk loop' $O/$n.log | cut -c1-260)"; md5sum /tmp/c3w/o_$n.gout.1 /tmp/c3w/o_$n.dpout.1 2>/dev/null | cut -c1-32 | tr '\n' ' '; echo; rm -f /tmp/c3w/o_$n*; }
SE150="soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -b 2"
SE100="soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r100.fq -L 100 -b 2"
run se150_post $SE150
run se150_post_2 $SE150
S3DP_POST_THREAD=0 run se150_nopost $SE150
run se100_post $SE100
S3DP_POST_THREAD=0 run se100_nopost $SE100
