#!/usr/bin/env python3
"""End-to-end wall clock of the drop-in command line: FASTQ in, SAM out ("reads/s (wall, SAM written)").

Builds a synthetic single-chromosome genome and its index FILES on the GPU (s3dp.builder, byte-identical to the reference tools'),
simulates reads into a FASTQ file, runs `soap3-dp single ...` on them and reports the phases the program prints.  The index files
go to --workdir (default: a temporary directory under /tmp); sizes: about 5.5 bytes per genome base."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=200_000_000)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mismatch", type=int, default=4)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--threads", type=int, default=0, help="host threads (0 = all)")
    ap.add_argument("--repeat-frac", type=float, default=0.0, help="share of the genome planted with GRCh38-like repeat families (bench.py's generator)")
    ap.add_argument("--devices", default="", help="-c argument of the command line (e.g. 0,0: two members on one GPU)")
    ap.add_argument("--bam", type=int, default=0, help="single reads: -b 3 (BAM output) instead of -b 2")
    ap.add_argument("--devnull", type=int, default=0, help="single reads: a second run with the SAM written to /dev/null")
    ap.add_argument("--default-mode", type=int, default=0, help="single reads: the reference's default mode -- BWT stage at Soap3MisMatchAllow mismatches, the reads it leaves through seeding + DP (.dpout.1)")
    ap.add_argument("--pairs", type=int, default=0, help="paired-end default mode (DP stages on) with this many 2 x read-len pairs instead of single reads")
    a = ap.parse_args()
    import torch
    import s3dp
    from s3dp import builder
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    work = a.workdir or tempfile.mkdtemp(prefix="s3dp_e2e_", dir="/tmp")
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(1)
    t = time.time()
    if a.repeat_frac > 0:
        sys.path.insert(0, ROOT)
        import bench
        codes = bench.make_genome(a.genome_bp, 1, torch.device(dev), a.repeat_frac)[0].cpu().numpy()
        torch.cuda.empty_cache()
    else:
        codes = rng.integers(0, 4, size=a.genome_bp, dtype=np.uint8)
    hi = builder.build_host_index(codes, device=dev)
    hi.write_files(os.path.join(work, "g.index"), with_lkt=True)
    t_index = time.time() - t
    # this process built the index on the GPU and is about to start the command line as a child: what torch's allocator still caches of the build (tens of
    # gigabytes) would be missing from the child's HBM -- it sizes its lookup tables by what is free
    del hi
    if dev != "cpu":
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        print("[bench_e2e] HBM free before the command line starts: %.1f of %.1f GB" % (free / 1e9, total / 1e9), file=sys.stderr)
    if a.pairs:
        return pairs_mode(a, codes, work, rng, t_index)
    # reads: uniform start, strand 50/50, 0-4 substitutions (55/25/12/5/3 %), quality 'I'
    t = time.time()
    L, N = a.read_len, a.reads
    st = rng.integers(0, a.genome_bp - L, size=N)
    idx = st[:, None] + np.arange(L)[None, :]
    seq = codes[idx]
    nsub = rng.choice(5, size=N, p=[0.55, 0.25, 0.12, 0.05, 0.03])
    for k in range(1, 5):
        rows = np.nonzero(nsub >= k)[0]
        cols = rng.integers(0, L, size=len(rows))
        seq[rows, cols] = (seq[rows, cols] + rng.integers(1, 4, size=len(rows))) % 4
    neg = rng.random(N) < 0.5
    seq[neg] = 3 - seq[neg][:, ::-1]
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[seq]
    fq = os.path.join(work, "reads.fq")
    with open(fq, "wb") as f:
        qual = b"I" * L
        block = 200_000
        for b0 in range(0, N, block):
            parts = []
            for i in range(b0, min(N, b0 + block)):
                parts.append(b"@r%d\n" % i + letters[i].tobytes() + b"\n+\n" + qual + b"\n")
            f.write(b"".join(parts))
    os.sync()                                  # (the index and read files were written seconds ago: with gigabytes of them still dirty in the page cache the
    t_reads = time.time() - t                  #  program's first pinned allocations wait for the write-back -- up to a second that is the harness's, not the program's)
    env = dict(os.environ)
    if a.threads:
        env["S3DP_HOST_THREADS"] = str(a.threads)
    exe = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")
    t = time.time()
    out = subprocess.run([exe, "single", os.path.join(work, "g.index"), fq, "-L", str(L)] + ([] if a.default_mode else ["-s", str(a.mismatch)]) + ["-b", "3" if a.bam else "2", "-o", os.path.join(work, "out")] +
                         (["-c", a.devices] if a.devices else []), capture_output=True, text=True, env=env)
    wall = time.time() - t
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-2000:], file=sys.stderr)
        raise SystemExit(1)
    if os.environ.get("S3DP_VERBOSE"):
        print(out.stdout, out.stderr, file=sys.stderr)
    ph = {k: float(v) for k, v in re.findall(r"\[Main\] (Loading time|Read load time|Alignment \+ output time|Total elapsed time) :\s+([0-9.]+)", out.stdout)}
    aligned = int(re.search(r"Overall number of reads aligned: (\d+)", out.stdout).group(1))
    sam_bytes = sum(os.path.getsize(os.path.join(work, f)) for f in os.listdir(work) if f.startswith("out.gout."))
    sl = re.search(r"\[Main\] Stream loop time : (.*)", out.stdout)
    ltc = re.search(r"Lookup tables in device memory : (\d+) characters", out.stdout)
    dv = re.search(r"\[Main\] Device time : (.*)", out.stdout)
    m = re.search(r"\[Main\] Read load time :\s+([0-9.]+) seconds \(loader", out.stdout)
    res = {"mode": "default (DP on)" if a.default_mode else "-s %d (DP off)" % a.mismatch, "output": "BAM (-b 3)" if a.bam else "SAM (-b 2)", "genome_bp": a.genome_bp, "repeat_frac": a.repeat_frac, "devices": a.devices or "0", "loader_thread_s": float(m.group(1)) if m else None, "reads": N, "read_len": L, "max_mismatch": a.mismatch, "host_threads": a.threads or os.cpu_count(),
           "index_build_and_write_s": t_index, "fastq_write_s": t_reads, "wall_s_total": wall, "phases_s": ph, "aligned_reads": aligned,
           "stream_loop": sl.group(1) if sl else None, "device_time": dv.group(1) if dv else None, "lookup_chars": int(ltc.group(1)) if ltc else None,
           "sam_bytes": sam_bytes, "fastq_bytes": os.path.getsize(fq),
           "reads_per_s_wall_incl_index_load": N / wall,
           # the command line streams: reading, aligning and writing overlap inside "Alignment + output time" (it starts when the index is resident)
           # (the loader runs ahead while the index is read; the pipeline cannot be faster than its slower stage, so the larger of the two times counts)
           "reads_per_s_reads_in_sam_out": N / (max(ph.get("Alignment + output time", 0), float(m.group(1)) if m else 0.0) or wall)}
    # the same run with the SAM going to /dev/null: what the program does when the file system does not set the pace
    if a.devnull:
        for f in os.listdir(work):
            if f.startswith("out.gout."):
                os.remove(os.path.join(work, f))
        for k in (1,) if not a.devices else range(1, len(a.devices.split(",")) + 1):
            os.symlink("/dev/null", os.path.join(work, "out.gout.%d" % k))
        t = time.time()
        out2 = subprocess.run([exe, "single", os.path.join(work, "g.index"), fq, "-L", str(L), "-s", str(a.mismatch), "-b", "2", "-o", os.path.join(work, "out")] +
                              (["-c", a.devices] if a.devices else []), capture_output=True, text=True, env=env)
        wall2 = time.time() - t
        if out2.returncode == 0:
            ph2 = {k: float(v) for k, v in re.findall(r"\[Main\] (Loading time|Read load time|Alignment \+ output time|Total elapsed time) :\s+([0-9.]+)", out2.stdout)}
            res["sam_to_dev_null"] = {"wall_s_total": wall2, "phases_s": ph2,
                                      "reads_per_s_reads_in_sam_out": N / max(ph2.get("Alignment + output time", wall2), ph2.get("Read load time", 0))}
    if not a.workdir:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(res))


def pairs_mode(a, codes, work, rng, t_index):
    """BASELINE.json configs[2] in small: 2 x L pairs, insert 350 +- 40, `soap3-dp pair -u 500 -v 1` with the DP stages on.  90 % of the pairs
    carry 0-2 substitutions per end (the BWT ladder pairs them), 6 % have a 3-base deletion or 2-base insertion in one end (default DP),
    3 % in both ends (deep DP), 1 % a random mate (single-end DP, .unpair)."""
    t = time.time()
    L, N = a.read_len, a.pairs
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    files = [os.path.join(work, "reads_%d.fq" % tag) for tag in (1, 2)]
    fh = [open(f, "wb") for f in files]
    qual = b"I" * L

    def subs(x):
        n = rng.choice(3, size=len(x), p=[0.7, 0.2, 0.1])
        for k in (1, 2):
            rows = np.nonzero(n >= k)[0]
            cols = rng.integers(0, L, size=len(rows))
            x[rows, cols] = (x[rows, cols] + rng.integers(1, 4, size=len(rows))) % 4

    def indel(x, rows):
        for r in rows:
            c = int(rng.integers(40, L - 40))
            if r & 1:
                x[r, c:L] = x[r, c + 3:L + 3]                                   # deletion of 3
            else:
                x[r, c + 2:L] = x[r, c:L - 2].copy(); x[r, c:c + 2] = rng.integers(0, 4, size=2)   # insertion of 2
    for b0 in range(0, N, 500_000):
        n = min(500_000, N - b0)
        ins = np.maximum(np.clip(rng.normal(350, 40, size=n).astype(np.int64), 0, 480), L + 10)
        st = rng.integers(0, a.genome_bp - 600, size=n)
        kind = rng.choice(4, size=n, p=[0.90, 0.06, 0.03, 0.01])
        left = codes[st[:, None] + np.arange(L + 4)[None, :]]
        right = codes[(st + ins - L)[:, None] + np.arange(L + 4)[None, :]]
        subs(left); subs(right)
        indel(right, np.nonzero(kind == 1)[0])
        indel(left, np.nonzero(kind == 2)[0]); indel(right, np.nonzero(kind == 2)[0])
        junk = np.nonzero(kind == 3)[0]
        right[junk] = rng.integers(0, 4, size=(len(junk), L + 4))
        a_ = left[:, :L]
        b_ = 3 - right[:, :L][:, ::-1]
        swap = rng.random(n) < 0.5
        for tag, rr in ((1, np.where(swap[:, None], b_, a_)), (2, np.where(swap[:, None], a_, b_))):
            txt = letters[rr]
            fh[tag - 1].write(b"".join(b"@p%d/%d\n" % (b0 + i, tag) + txt[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(n)))
    for f in fh:
        f.close()
    os.sync()                                  # (the index and read files were written seconds ago: with gigabytes of them still dirty in the page cache the
    t_reads = time.time() - t                  #  program's first pinned allocations wait for the write-back -- up to a second that is the harness's, not the program's)
    env = dict(os.environ)
    if a.threads:
        env["S3DP_HOST_THREADS"] = str(a.threads)
    exe = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")
    t = time.time()
    out = subprocess.run([exe, "pair", os.path.join(work, "g.index"), files[0], files[1], "-L", str(L), "-u", "500", "-v", "1", "-o", os.path.join(work, "out")],
                         capture_output=True, text=True, env=env)
    wall = time.time() - t
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-2000:], file=sys.stderr)
        raise SystemExit(1)
    if os.environ.get("S3DP_VERBOSE"):
        print(out.stdout, out.stderr, file=sys.stderr)
    ph = {k: float(v) for k, v in re.findall(r"\[Main\] (Loading time|Read load time|Alignment \+ output time|Total elapsed time|DP stages time) :\s+([0-9.]+)", out.stdout)}
    ch = re.search(r"\[Main\] Chunk loop time : (.*)", out.stdout)
    dv = re.search(r"\[Main\] Device time : (.*)", out.stdout)
    counts = {k: int(v) for k, v in re.findall(r"\[Main\] (Number of pairs aligned by new default DP|Number of pairs aligned by DP|Number of pairs aligned by deep DP|"
                                               r"Pairs left without a proper pairing|Overall number of reads aligned): (\d+)", out.stdout)}
    stage = re.search(r"DP stages time :\s+[0-9.]+ seconds \((.*)\)", out.stdout)
    res = {"mode": "pair, DP stages on", "dp_stage_seconds": stage.group(1) if stage else None, "genome_bp": a.genome_bp, "pairs": N, "read_len": L, "host_threads": a.threads or os.cpu_count(),
           "pair_mix": {"bwt": 0.90, "one_end_indel": 0.06, "both_ends_indel": 0.03, "random_mate": 0.01},
           "chunk_loop": ch.group(1) if ch else None, "device_time": dv.group(1) if dv else None,
           "index_build_and_write_s": t_index, "fastq_write_s": t_reads, "wall_s_total": wall, "phases_s": ph, "counts": counts,
           "sam_bytes": sum(os.path.getsize(os.path.join(work, "out." + e)) for e in ("gout.1", "dpout.1", "unpair")),
           "pairs_per_s_wall_incl_index_load": N / wall,
           "pairs_per_s_reads_in_sam_out": N / (ph.get("Read load time", 0) + ph.get("Alignment + output time", 0) or wall)}
    if not a.workdir:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
