#!/bin/bash
# round 6 (finding: no -- the next process on the device starts 2 s slower; the switch S3DP_EXIT_FAST this script used is gone again): does a process that ends without freeing its 219 GB index give the memory back as fast as one that frees it?  Wall clock of the single-end command
# line twice back to back in each mode (the second run must find room for both flank tables).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06x2}; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp 3100000000 --reads ${READS:-10000000} --repeat-frac 0.5 --workdir $W > $O/e2e.json 2> $O/e2e.log
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
CLI="soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out"
for mode in 0 1 0 1; do
  for k in 1 2; do
    t0=$(date +%s.%N); S3DP_EXIT_FAST=$mode $CLI > $O/run_${mode}_$k.txt 2>&1; rc=$?; t1=$(date +%s.%N)
    echo "exit_fast=$mode run $k rc=$rc wall $(echo "$t1 - $t0" | bc) s; $(grep -o 'Total elapsed time : *[0-9.]*' $O/run_${mode}_$k.txt); $(grep -o 'Flank tables.*' $O/run_${mode}_$k.txt)"
  done
done
rm -rf $W
