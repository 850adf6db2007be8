#!/usr/bin/env python3
"""Compare the all-best ladder with and without the second search pass (S3DP_HEAVY=0 walks every read with one lane) on the bench's repeat-rich
workload at a chosen scale, several runs each; prints the reads whose results differ.  Debugging aid for the heavy-read path (GPU only)."""
import argparse
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=200_000_000)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--repeat-frac", type=float, default=0.5)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--index-dir", default=None)
    ap.add_argument("--settings", default="1024:4096,1024:1000000000")
    ap.add_argument("--heavy-first", type=int, default=0, help="1: the batch's first ladder (the one that grows its buffers) runs with the first setting")
    ap.add_argument("--counters", type=int, default=0)
    a = ap.parse_args()
    import torch
    import bench
    import s3dp
    from s3dp import builder
    dev = torch.device("cuda:0")
    codes, _ = bench.make_genome(a.genome_bp, 1, dev, a.repeat_frac)
    reads2d, tpos, tstrand, tsub = bench.make_reads(codes, a.reads, 100, 2026, dev)
    wpq = s3dp.word_per_query(100)
    q = bench.pack_on_device(reads2d, wpq)
    lens = np.full(a.reads, 100, dtype=np.uint32)
    tsub = tsub.cpu().numpy()
    prefix = os.path.join(a.index_dir, "bench_repeat-rich_%d.index" % a.genome_bp) if a.index_dir else None
    if prefix and os.path.exists(prefix + ".sa"):
        hi = s3dp.HostIndex.from_files(prefix)
    else:
        hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=False)
    torch.cuda.empty_cache()
    ix = s3dp.Index(hi, device=0)
    b = s3dp.Batch(ix, a.reads, wpq)
    b.upload(q, lens)

    def run():
        b.align_single_allbest(4, 1000)
        b.sync()
        phase, werr, offs, pos, strand, mis = b.fetch_alignments()
        return phase.copy(), offs.copy(), pos.copy(), strand.copy(), mis.copy()

    first = None
    if a.counters:
        b.enable_counters(True)
    if a.heavy_first:
        h, r = a.settings.split(",")[0].split(":")
        os.environ["S3DP_HEAVY"] = h
        os.environ["S3DP_HEAVY_RANK"] = r
        first = run()
    os.environ["S3DP_HEAVY"] = "0"
    ref = run()
    if first is not None:
        cnt_ref = np.diff(ref[1].astype(np.int64))
        cnt_got = np.diff(first[1].astype(np.int64))
        bad = np.nonzero((ref[0] != first[0]) | (cnt_ref != cnt_got))[0]
        print("first ladder of the batch (heavy): %d reads differ" % bad.size, flush=True)
        for i in bad[:12]:
            print("   read %d planted %d: phase %d -> %d, occurrences %d -> %d" % (i, int(tsub[i]), ref[0][i], first[0][i], cnt_ref[i], cnt_got[i]))
    print("reference (S3DP_HEAVY=0): aligned %d, occurrences %d" % (int((ref[0] > 0).sum()), int(ref[1][-1])), flush=True)
    for setting in a.settings.split(","):
        h, r = setting.split(":")
        os.environ["S3DP_HEAVY"] = h
        os.environ["S3DP_HEAVY_RANK"] = r
        for k in range(a.runs):
            got = run()
            cnt_ref = np.diff(ref[1].astype(np.int64))
            cnt_got = np.diff(got[1].astype(np.int64))
            bad = np.nonzero((ref[0] != got[0]) | (cnt_ref != cnt_got))[0]
            same_lists = bad.size == 0 and np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3]) and np.array_equal(ref[4], got[4])
            print("heavy=%s rank=%s run %d: %d reads differ in phase / count; lists identical: %s" % (h, r, k, bad.size, same_lists), flush=True)
            for i in bad[:12]:
                print("   read %d planted %d: phase %d -> %d, occurrences %d -> %d" % (i, int(tsub[i]), ref[0][i], got[0][i], cnt_ref[i], cnt_got[i]))
            for i in bad[:4]:
                if cnt_ref[i] <= 24 and cnt_got[i] <= 24:
                    for nm, r in (("ref", ref), ("got", got)):
                        a0, a1 = int(r[1][i]), int(r[1][i + 1])
                        print("      %s: %s" % (nm, " ".join("%d:%d/%d" % (r[2][k], int(r[3][k]), r[4][k]) for k in range(a0, a1))))


if __name__ == "__main__":
    main()
