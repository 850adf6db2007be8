#!/bin/bash
# round-3 measurement set, part 1 (run through gpurun from the repository root): kernel statistics and counters of the bench ladder on the repeat-rich and the
# uniform genome, profiles/pmc_traffic.json's source, the command line end to end, the DP kernel.  Part 2 is the default `python bench.py` with the new
# pmc_traffic.json committed.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export S3DP_PROFILE_OUT=gpurun_out/r03
mkdir -p $S3DP_PROFILE_OUT
tools/run_profiles.sh stats pmc dp 2>&1 | tail -30
S3DP_PROFILE_OUT=gpurun_out/r03/uniform S3DP_PROFILE_REPEAT_FRAC=0.0 tools/run_profiles.sh stats pmc 2>&1 | tail -12
python3 tools/make_pmc_traffic.py "$(cat .commit 2>/dev/null || echo unknown)" 0.5=/tmp/pmc_fetch_0.5.csv 0.0=/tmp/pmc_fetch_0.0.csv > gpurun_out/r03/pmc_traffic.json; head -c 600 gpurun_out/r03/pmc_traffic.json; echo
O=gpurun_out/r03
S3DP_VERBOSE=1 timeout 900 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 > $O/e2e_se_uniform.json 2> $O/e2e_se_uniform.log; echo "e2e uniform rc=$?"; cut -c1-700 $O/e2e_se_uniform.json
S3DP_VERBOSE=1 timeout 1200 python3 tools/bench_e2e.py --genome-bp 3100000000 --reads 20000000 --repeat-frac 0.5 > $O/e2e_se_repeat.json 2> $O/e2e_se_repeat.log; echo "e2e repeat rc=$?"; cut -c1-700 $O/e2e_se_repeat.json
timeout 900 python3 tools/bench_e2e.py --genome-bp 1000000000 --reads 20000000 > $O/e2e_se_uniform_1g.json 2> /dev/null; cut -c1-500 $O/e2e_se_uniform_1g.json
S3DP_VERBOSE=1 timeout 1200 python3 tools/bench_e2e.py --genome-bp 1000000000 --pairs 10000000 --read-len 150 > $O/e2e_pairs_dp.json 2> $O/e2e_pairs_dp.log; echo "pairs rc=$?"; cut -c1-900 $O/e2e_pairs_dp.json
grep -h "deep DP stage\|member 0" $O/e2e_pairs_dp.log | tail -8 | cut -c1-300
du -sh $O
