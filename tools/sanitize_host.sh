#!/bin/bash
# AddressSanitizer + UBSan over the host side (read loaders, SAM writers, DP record writers) and the oracle, CPU build only
# (GPU sanitizers are not available on the pool).  Runs the CPU tools of tests/tools on the golden fixtures; prints one line per run.
# libs3dp.so must be built (make -C soap3-dp_amd); nothing here calls a device function.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/soap3-dp_amd/host; L=$ROOT/soap3-dp_amd; G=$ROOT/tests/golden
W=${1:-/tmp/s3dp_sanitize}; mkdir -p "$W"; cd "$W" || exit 1
SAN="-O1 -g -mssse3 -fsanitize=address,undefined -fno-omit-frame-pointer"
LINK="-L$L -ls3dp -lz -lm -Wl,-rpath,$L"
gcc $SAN -c "$ROOT/oracle/s3_oracle.c" -o o1.o && gcc $SAN -c "$ROOT/oracle/s3_oracle_pe.c" -o o2.o || exit 1
g++ $SAN -std=c++17 -pthread -o oracle_sam "$ROOT/tests/tools/oracle_sam.cpp" "$H/s3_host.cpp" "$H/s3_bam.cpp" o1.o o2.o $LINK || exit 1
g++ $SAN -std=c++17 -pthread -o dp_sam "$ROOT/tests/tools/dp_sam.cpp" "$H/s3_host.cpp" "$H/s3_bam.cpp" "$H/s3_host_dp.cpp" $LINK || exit 1
g++ $SAN -std=c++17 -pthread -o dp_pair_sam "$ROOT/tests/tools/dp_pair_sam.cpp" "$H/s3_host.cpp" "$H/s3_bam.cpp" "$H/s3_host_dp.cpp" "$H/s3_host_dp_pair.cpp" $LINK || exit 1
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
cp "$G/g1/reads.fq" g1.fq; cp "$G/g6/reads.fq" g6.fq; cp "$G/g3/reads.fq" g3.fq; cp "$G/g4/reads_1.fq" g4_1.fq; cp "$G/g4/reads_2.fq" g4_2.fq
cp "$G/dp1/reads.fq" dp1.fq; cp "$G/dp2/reads_1.fq" dp2_1.fq; cp "$G/dp2/reads_2.fq" dp2_2.fq
bad=0
run() { local name=$1; shift; timeout 900 "$@" > "$name.log" 2>&1 < /dev/null; local rc=$?
        local n; n=$(grep -c 'runtime error\|ERROR: AddressSanitizer' "$name.log"); echo "$name rc=$rc findings=$n"; [ $rc -ne 0 ] || [ "$n" -ne 0 ] && bad=1; }
run g1_s4 ./oracle_sam single "$G/g1/index" g1.fq -L 100 -s 4 -o o1
run g6_s4 ./oracle_sam single "$G/g6/index" g6.fq -L 100 -s 4 -o o6
run g3_long ./oracle_sam single "$G/g3/index" g3.fq -L 200 -s 3 -h 3 -o o3
run g4_pair ./oracle_sam pair "$G/g4/index" g4_1.fq g4_2.fq -L 100 -u 500 -v 200 -s 3 -h 3 -o o4
run dp1 ./dp_sam "$G/dp1/index" dp1.fq "$G/dp1/results.txt" d1 -L 100 -h 1 -p
for o in "" "-h 1 -p" "-h 4 -p" "-p -B 0" "-h 3"; do
    run "dp2_$(echo $o | tr -d ' -')" ./dp_pair_sam "$G/dp1/index" dp2_1.fq dp2_2.fq "$G/dp2/results.txt" "$G/dp2/stats.txt" d2 -L 100 $o
done
# the read loader on its own (line-by-line FASTQ path, SIMD packing, recycled read sets, BAM input): chunked dumps of plain, odd and BAM files
g++ $SAN -std=c++17 -pthread -o chunk_reads "$ROOT/tests/tools/chunk_reads.cpp" "$H/s3_host.cpp" "$H/s3_bam.cpp" "$H/s3_host_dp.cpp" "$H/s3_host_dp_pair.cpp" $LINK || exit 1
printf '@a\nACGTACGTACGTACGTACGT+ACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIII\n@b/1 x\nACGNNacgtu\n+b\nII@IIIIIIIII\n@c\nAC\nGT\n+\nIIII\n@d\r\nACGT\r\n+\r\nIIII\r\n@e\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nII' > odd.fq
run loader_g4_pairs ./chunk_reads 333 100 ld1.txt g4_1.fq g4_2.fq
run loader_g1 env S3DP_HOST_THREADS=5 ./chunk_reads 257 100 ld2.txt g1.fq
run loader_odd ./chunk_reads 2 16 ld3.txt odd.fq
run loader_odd_generic env S3DP_GENERIC_PARSER=1 ./chunk_reads 2 16 ld4.txt odd.fq
cmp ld3.txt ld4.txt || { echo "loader_odd: the two parsers differ"; bad=1; }
run loader_bam ./chunk_reads 100 120 ld5.txt "bam:$G/bam/se.bam"
run loader_bam_pairs ./chunk_reads 100 100 ld6.txt "bampair:$G/bam/pe.bam"
# ThreadSanitizer over the two-thread pair loader and the record formatting
TS="-O1 -g -mssse3 -fsanitize=thread"
gcc $TS -c "$ROOT/oracle/s3_oracle.c" -o t1.o && gcc $TS -c "$ROOT/oracle/s3_oracle_pe.c" -o t2.o &&
g++ $TS -std=c++17 -pthread -o oracle_sam_tsan "$ROOT/tests/tools/oracle_sam.cpp" "$H/s3_host.cpp" "$H/s3_bam.cpp" t1.o t2.o $LINK || exit 1
timeout 900 ./oracle_sam_tsan pair "$G/g4/index" g4_1.fq g4_2.fq -L 100 -u 500 -v 200 -s 3 -h 3 -o t4 > tsan_g4.log 2>&1 < /dev/null; rc=$?
n=$(grep -c 'WARNING: ThreadSanitizer' tsan_g4.log); echo "tsan_g4_pair rc=$rc findings=$n"; [ $rc -ne 0 ] || [ "$n" -ne 0 ] && bad=1
exit $bad
