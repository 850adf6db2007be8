#!/bin/bash
# the command line on one set of files under several settings of the host side (run through gpurun from the repository root):
# tools/e2e_variants.sh [genome_bp] [reads]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
G=${1:-1000000000}; N=${2:-20000000}
O=gpurun_out/r03b; mkdir -p $O
W=/tmp/e2e_w; rm -rf $W
python3 tools/bench_e2e.py --genome-bp $G --reads $N --workdir $W > $O/e2e_base.json 2> $O/e2e_base.log; cut -c1-900 $O/e2e_base.json
run() {
  local tag=$1; shift
  env "$@" soap3-dp_amd/soap3-dp single $W/g.index $W/reads.fq -L 100 -s 4 -b 2 -o $W/out 2>&1 | grep "Stream loop\|Alignment + output\|Read load" | sed "s/^/$tag: /"
}
run base A=1
run base2 A=1
run blocking S3DP_BLOCKING_SYNC=1
for t in 8 12 20 32; do run threads$t S3DP_HOST_THREADS=$t; done
run chunk512k S3DP_CHUNK_READS=524288
run chunk2m S3DP_CHUNK_READS=2097152
rm -f $W/out.gout.1; ln -s /dev/null $W/out.gout.1
run devnull A=1
rm -f $W/out.gout.1
