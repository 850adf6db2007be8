#!/bin/bash
# round 6: the timeline of the single-end command line (FASTQ in, SAM to /dev/null; repeat-rich 3.1 Gbp genome, READS reads): plain runs in the settings
# given as arguments ("VAR=val,VAR=val" each), then the first setting under rocprofv3 --kernel-trace and tools/trace_gaps.py on the window of the search
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06x}; shift; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W /tmp/prof_se
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads ${READS:-20000000} --repeat-frac 0.5 --workdir $W > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; cut -c1-300 $O/e2e_se_repeat.json; echo
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
CLI="soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out"
for cfg in ${PLAIN:-"$@"}; do
  envs=""; [ "$cfg" != base ] && envs=$(echo "$cfg" | tr ',' ' ')
  tag=$(echo $cfg | tr ',=' '__')
  for k in 1 2; do
    env $envs $CLI > $O/run_${tag}_$k.txt 2>&1
    grep "Stream loop\|Alignment + output\|Device time" $O/run_${tag}_$k.txt | sed "s/^/$tag $k: /" | cut -c1-300
  done
done
S3DP_VERBOSE=1 $CLI 2>&1 | grep "s3dp\]" | grep " MB " > $O/allocations.txt; echo "allocations and growths of 1 MB and more in a plain run: $(wc -l < $O/allocations.txt), $(awk '{s+=$(NF-1)} END {print s}' $O/allocations.txt) ms"; tail -40 $O/allocations.txt
cfg=${1:-base}; [ "$cfg" != base ] && export $(echo "$cfg" | tr ',' ' ')
timeout 400 rocprofv3 --kernel-trace ${HIPTRACE:+--hip-trace} --output-format csv -d /tmp/prof_se -o cli -- $CLI > $O/prof.log 2>&1; echo "prof rc=$? $(grep 'Alignment + output' $O/prof.log)"
f=$(find /tmp/prof_se -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_gaps.py $f from firstKernel | tee $O/trace_gaps.txt
a=$(find /tmp/prof_se -name "*hip_api_trace.csv" | head -1)
[ -n "$a" ] && python3 tools/trace_api_stalls.py $a $f firstKernel 8 | tee $O/trace_api_stalls.txt
rm -rf $W /tmp/prof_se
