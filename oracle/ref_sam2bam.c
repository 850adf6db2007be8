/* ref_sam2bam.c -- test infrastructure: writes SAM text (with header) as a BAM file with the REFERENCE's samtools-0.1.18 (objects built by oracle/ref.mk
 * from /root/reference/samtools-0.1.18; samopen "r" / "wb", the calls the reference's own BAM output goes through, SOAP3-DP.cu:413).  Used by
 * tests/golden/make_fixtures.sh to make the BAM files the product's `-bam` input is tested on (tests/golden/bam/): a reader checked only against its own
 * writer's files would be a self-comparison.
 *   ref_sam2bam in.sam out.bam */
#include <stdio.h>
#include "sam.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_sam2bam in.sam out.bam\n"); return 2; }
    samfile_t *in = samopen(argv[1], "r", 0);
    if (!in || !in->header) { fprintf(stderr, "cannot open %s as SAM with a header\n", argv[1]); return 1; }
    samfile_t *out = samopen(argv[2], "wb", in->header);
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    bam1_t *b = bam_init1();
    long n = 0;
    int r;
    while ((r = samread(in, b)) >= 0) { samwrite(out, b); n++; }
    bam_destroy1(b);
    samclose(out);
    samclose(in);
    if (r < -1) { fprintf(stderr, "bad record after %ld records\n", n); return 1; }
    fprintf(stderr, "%ld records\n", n);
    return 0;
}
