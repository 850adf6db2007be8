/* ref_bam2sam.c -- test infrastructure: reads a BAM file with the REFERENCE's samtools-0.1.18 (objects built by oracle/ref.mk from
 * /root/reference/samtools-0.1.18) and prints it as SAM text with the header, the way the reference's own SAM output is written (samopen "wh").
 * Used by tests/test_host_bam_cpu.py to check that what `soap3-dp -b 3` writes is the same set of records as its SAM output.
 *   ref_bam2sam in.bam > out.sam */
#include <stdio.h>
#include "sam.h"

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_bam2sam in.bam\n"); return 2; }
    samfile_t *in = samopen(argv[1], "rb", 0);
    if (!in) { fprintf(stderr, "cannot open %s as BAM\n", argv[1]); return 1; }
    samfile_t *out = samopen("-", "wh", in->header);
    if (!out) return 1;
    bam1_t *b = bam_init1();
    long n = 0;
    int r;
    while ((r = samread(in, b)) >= 0) { samwrite(out, b); n++; }
    bam_destroy1(b);
    samclose(out);
    samclose(in);
    if (r < -1) { fprintf(stderr, "truncated or corrupt BAM after %ld records\n", n); return 1; }
    return 0;
}
