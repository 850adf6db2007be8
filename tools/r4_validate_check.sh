#!/bin/bash
# round 4: the wave-shared verification of long reads -- parity tests, then the command line on 150-base single reads in the default mode
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O /tmp/s3dp_ix
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_sam.py tests/test_gpu_cli_vs_oracle_scale.py -m gpu -x -q -k "long" > $O/pytest_long.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_long.log)"
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
rm -f /tmp/c3w/out.*
SE="soap3-dp_amd/soap3-dp single /tmp/c3w/g.index /tmp/c3w/r1.fq -L 150 -b 2"
export S3DP_VERBOSE=1
sync; timeout 200 $SE -o /tmp/c3w/se_plain > $O/se_plain.log 2>&1; echo "plain rc=$? $(grep 'Alignment + output\|Device time\|Chunk loop' $O/se_plain.log)"
rm -f /tmp/c3w/se_plain*
sync; timeout 200 $SE -s 2 -o /tmp/c3w/se_nodp > $O/se_nodp.log 2>&1; echo "DP off (-s 2) rc=$? $(grep 'Alignment + output\|Device time' $O/se_nodp.log)"
rm -f /tmp/c3w/se_nodp*
rm -rf /tmp/prof_se
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_se -o se -- $SE -o /tmp/c3w/se_prof > $O/se_prof.log 2>&1; echo "prof rc=$?"
f=$(find /tmp/prof_se -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/se150_dp_kernel_stats.csv && head -12 $O/se150_dp_kernel_stats.csv | cut -c1-180
