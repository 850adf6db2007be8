#!/bin/bash
# round 4: does the paired command line finish under rocprofv3 --kernel-trace, and how much does its time vary from run to run on one box?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O /tmp/s3dp_ix
timeout 300 python3 bench.py --index-dir /tmp/s3dp_ix --build-only 1 > /dev/null 2> $O/index_build.log; echo "index rc=$?"
timeout 500 python3 tools/bench_c3.py --pairs 10000000 --index-dir /tmp/s3dp_ix --workdir /tmp/c3w > $O/c3_10m.json 2> $O/c3_10m.log; echo "c3 rc=$?"
grep -o "\"Alignment + output time\": [0-9.]*" $O/c3_10m.json
head -n 4000000 /tmp/c3w/r1.fq > /tmp/c3w/s1.fq; head -n 4000000 /tmp/c3w/r2.fq > /tmp/c3w/s2.fq
CLI="soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index"
OPT="-L 150 -u 500 -v 1 -b 2"
for k in 1 2 3; do timeout 120 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/plain_$k.log 2>&1; echo "plain $k rc=$? $(grep 'Alignment + output' $O/plain_$k.log)"; done
S3DP_POST_THREAD=0 timeout 120 $CLI /tmp/c3w/r1.fq /tmp/c3w/r2.fq $OPT -o /tmp/c3w/outp > $O/nopost.log 2>&1; echo "no post thread rc=$? $(grep 'Alignment + output' $O/nopost.log)"
run_prof() {  # name, files..., env through the caller
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o cli -- $CLI "$@" $OPT -o /tmp/c3w/outq > $O/prof_$name.log 2>&1 &
  local pid=$! t=0
  while kill -0 $pid 2>/dev/null && [ $t -lt ${LIMIT:-150} ]; do sleep 2; t=$((t+2)); done
  if kill -0 $pid 2>/dev/null; then
    echo "prof $name: still running after $t s"
    app=$(pgrep -x soap3-dp | head -1)
    if [ -n "$app" ]; then
      for tk in /proc/$app/task/*; do echo "$(basename $tk) $(cat $tk/comm) state=$(grep State $tk/status | cut -f2) wchan=$(cat $tk/wchan 2>/dev/null)"; done > $O/hang_$name.tasks 2>&1
      timeout 60 rocgdb -p $app -batch -ex "thread apply all bt 12" > $O/hang_$name.bt 2>&1
      kill -9 $app
    fi
    sleep 2; kill -9 $pid 2>/dev/null
  else
    wait $pid; echo "prof $name rc=$? after $t s $(grep 'Alignment + output' $O/prof_$name.log)"
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  fi
}
LIMIT=90 run_prof small /tmp/c3w/s1.fq /tmp/c3w/s2.fq
LIMIT=150 run_prof full /tmp/c3w/r1.fq /tmp/c3w/r2.fq
export S3DP_POST_THREAD=0
LIMIT=150 run_prof full_nopost /tmp/c3w/r1.fq /tmp/c3w/r2.fq
ls $O
