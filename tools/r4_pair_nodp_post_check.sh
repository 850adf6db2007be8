#!/bin/bash
# round 4: pairs with the DP stages off through the post thread -- the command line's SAM parity tests, then 1 M pairs on a 200 Mbp genome with and without it (same files?)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
timeout 100 python3 -m pytest tests/test_gpu_cli_sam.py -m gpu -x -q > $O/pytest_cli_sam.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_cli_sam.log)"
timeout 60 python3 tools/bench_c3.py --genome-bp 200000000 --pairs 1000000 --workdir /tmp/c3w > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
for v in on off; do
  [ $v = off ] && export S3DP_POST_THREAD=0
  timeout 40 soap3-dp_amd/soap3-dp pair /tmp/c3w/g.index /tmp/c3w/r1.fq /tmp/c3w/r2.fq -L 150 -u 500 -v 1 -s 2 -b 2 -o /tmp/c3w/o_$v > $O/nodp_$v.log 2>&1
  echo "post $v rc=$? $(grep 'Alignment + output' $O/nodp_$v.log | sed 's/.*: *//') $(md5sum /tmp/c3w/o_$v.gout.1 | cut -c1-32) $(grep 'Chunk loop' $O/nodp_$v.log | cut -c1-150)"
done
