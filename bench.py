#!/usr/bin/env python3
"""bench.py -- reads/s of the alignment hot path (BASELINE.json configs[1]): synthetic genome of GRCh38
scale with a full-SA 2BWT + GPU-2BWT index resident in HBM, synthetic 100 bp single-end reads,
<= 4 mismatches, all-best (soap3-dp single ... -s 4 -h 2 -b 2), DP off.

A "step" = one pass of the whole single-end ladder (phases 1..4 mismatches: BWT search with lookup-table
start and check-and-extend, ordering, per-read merge, SA decode) over one resident batch of reads.
One process per GPU; reads are sharded (weak scaling), the index is replicated per GPU; the only
collective is the final reduce of the counters.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_genome(n, seed, device, repeat_frac=0.0):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    codes = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        codes[a:b] = torch.randint(0, 4, (b - a,), generator=g, device=device, dtype=torch.uint8)
    if repeat_frac > 0:
        # planted repeat families: 300..3000 bp units copied with ~1 % divergence
        cpu = torch.Generator()
        cpu.manual_seed(seed + 1)
        target = int(n * repeat_frac)
        placed = 0
        while placed < target:
            ulen = int(torch.randint(300, 3000, (1,), generator=cpu))
            copies = int(torch.randint(5, 200, (1,), generator=cpu))
            src = int(torch.randint(0, n - ulen, (1,), generator=cpu))
            unit = codes[src:src + ulen].clone()
            dst = torch.randint(0, n - ulen, (copies,), generator=cpu)
            for d in dst.tolist():
                u = unit.clone()
                nsub = max(1, ulen // 100)
                p = torch.randint(0, ulen, (nsub,), generator=cpu).to(device)
                u[p] = (u[p] + torch.randint(1, 4, (nsub,), generator=cpu).to(device).to(torch.uint8)) % 4
                codes[d:d + ulen] = u
            placed += ulen * copies
    return codes


def make_reads(codes, n_reads, L, seed, device, mix=(0.55, 0.25, 0.12, 0.05, 0.03)):
    """uniform start, strand Bernoulli(0.5), substitutions per read ~ Categorical(mix)"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = codes.numel()
    out = torch.empty((n_reads, L), dtype=torch.uint8, device=device)
    truth_pos = torch.empty(n_reads, dtype=torch.int64, device=device)
    truth_strand = torch.empty(n_reads, dtype=torch.uint8, device=device)
    truth_sub = torch.empty(n_reads, dtype=torch.uint8, device=device)
    ar = torch.arange(L, device=device)
    probs = torch.tensor(mix, device=device, dtype=torch.float32)
    chunk = 1 << 20
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        m = b - a
        start = torch.randint(0, n - L, (m,), generator=g, device=device)
        seg = codes[start[:, None] + ar[None, :]]
        nsub = torch.multinomial(probs, m, replacement=True, generator=g)
        score = torch.rand((m, L), generator=g, device=device)
        kth = torch.topk(score, len(mix) - 1, dim=1).indices                  # up to 4 distinct positions
        delta = torch.randint(1, 4, (m, len(mix) - 1), generator=g, device=device, dtype=torch.uint8)
        for j in range(len(mix) - 1):
            sel = nsub > j
            rows = torch.nonzero(sel).squeeze(1)
            cols = kth[rows, j]
            seg[rows, cols] = (seg[rows, cols] + delta[rows, j]) % 4
        strand = torch.randint(0, 2, (m,), generator=g, device=device, dtype=torch.uint8)
        rc = 3 - torch.flip(seg, [1])
        seg = torch.where(strand[:, None] == 1, rc, seg)
        out[a:b] = seg
        truth_pos[a:b] = start
        truth_strand[a:b] = strand
        truth_sub[a:b] = nsub.to(torch.uint8)
    return out, truth_pos, truth_strand, truth_sub


def pack_on_device(codes2d, wpq):
    """reference packed layout (32-interleaved words, chars LSB first) built on the device"""
    import torch
    n, L = codes2d.shape
    dev = codes2d.device
    round_up = (n + 31) // 32 * 32
    buf = torch.zeros((round_up, wpq * 16), dtype=torch.int64, device=dev)
    buf[:n, :L] = codes2d.to(torch.int64)
    sh = 2 * torch.arange(16, dtype=torch.int64, device=dev)
    words = (buf.view(round_up, wpq, 16) << sh).sum(dim=2)
    q = words.view(round_up // 32, 32, wpq).permute(0, 2, 1).reshape(-1)
    return (q & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


def make_pairs(codes, n_pairs, L, seed, device):
    """2 x L pairs, insert ~ N(350, 40) clipped to [L + 1, 500], '+/-' orientation, 0-4 substitutions per end"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = codes.numel()
    ins = torch.clamp(torch.round(torch.randn(n_pairs, generator=g, device=device) * 40 + 350), L + 1, 500).to(torch.int64)
    start = torch.randint(0, n - 600, (n_pairs,), generator=g, device=device)
    ar = torch.arange(L, device=device)
    a = codes[start[:, None] + ar[None, :]]
    bb = codes[(start + ins - L)[:, None] + ar[None, :]]
    probs = torch.tensor((0.55, 0.25, 0.12, 0.05, 0.03), device=device)
    for seg in (a, bb):
        nsub = torch.multinomial(probs, n_pairs, replacement=True, generator=g)
        kth = torch.topk(torch.rand((n_pairs, L), generator=g, device=device), 4, dim=1).indices
        delta = torch.randint(1, 4, (n_pairs, 4), generator=g, device=device, dtype=torch.uint8)
        for j in range(4):
            rows = torch.nonzero(nsub > j).squeeze(1)
            seg[rows, kth[rows, j]] = (seg[rows, kth[rows, j]] + delta[rows, j]) % 4
    bb = 3 - torch.flip(bb, [1])
    swap = torch.rand(n_pairs, generator=g, device=device) < 0.5
    first = torch.where(swap[:, None], bb, a)
    second = torch.where(swap[:, None], a, bb)
    out = torch.empty((2 * n_pairs, L), dtype=torch.uint8, device=device)
    out[0::2] = first
    out[1::2] = second
    return out


def extras(s3dp, ix, codes, dev, rank):
    """secondary figures (not the headline metric): BASELINE configs[2] shapes -- 2 x 150 bp pairs through the paired-end BWT stage, and the
    semi-global DP kernel on default-DP windows"""
    import torch
    res = {}
    L, n_pairs = 150, 2_000_000
    reads2d = make_pairs(codes, n_pairs, L, 4242, dev)
    wpq = s3dp.word_per_query(L)
    q = pack_on_device(reads2d, wpq)
    del reads2d
    lens = np.full(2 * n_pairs, L, dtype=np.uint32)
    b = s3dp.Batch(ix, 2 * n_pairs, wpq)
    b.upload(q, lens)
    b.align_pair(3, 8000, 2, 1, 500, 1000)
    b.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        b.align_pair(3, 8000, 2, 1, 500, 1000)
        b.sync()
    dt = (time.perf_counter() - t0) / 2
    pairs = b.fetch_pairs()[0]
    res["paired_end_2x150_bwt_stage"] = {"pairs": n_pairs, "ms_per_pass": dt * 1e3, "M_pairs_per_s": n_pairs / dt / 1e6,
                                          "properly_paired": int((pairs["phase"] > 0).sum()),
                                          "options": "pair -L 150 -s 3 -h 2 -u 500 -v 1 (DP stages off)"}
    b.close()
    # DP kernel: 150 bp reads against 648-column windows cut from the genome at the reads' places (default-DP shape), device time
    n_dp = 32768
    rng = np.random.default_rng(3)
    occ = np.zeros(n_dp, dtype=s3dp.OCC_DTYPE)
    reads2d = make_pairs(codes, n_dp, L, 777, dev)
    q = pack_on_device(reads2d, wpq)
    lens = np.full(2 * n_dp, L, dtype=np.uint32)
    bb = s3dp.Batch(ix, 2 * n_dp, wpq)
    bb.upload(q, lens)
    bb.align_pair(3, 8000, 2, 1, 500, 1000)
    pr = bb.fetch_pairs()[0]
    k = 0
    for p in range(n_dp):                                              # the first end's hit of every properly paired pair anchors a DP of the second end
        if pr["phase"][p] and k < n_dp:
            occ[k] = (2 * p, pr["bestA1"][p], pr["bestS1"][p], pr["bestM1"][p], (0, 0))
            k += 1
    dp = s3dp.DpAligner(n_dp, 152, 500 - 1 + 152 + 1, scheme=1)
    dp.half_end(bb, lens, occ[:k], 1, 500, (45, 45), 49, 49)
    t0 = time.perf_counter()
    resd, cig = dp.half_end(bb, lens, occ[:k], 1, 500, (45, 45), 49, 49)
    wall = time.perf_counter() - t0
    ms = dp.timing()
    cells = float(k) * L * 499.0
    res["default_dp_half_end"] = {"problems": k, "read_len": L, "window": 499, "kernel_ms": ms, "GCUPS_kernel": cells / (ms / 1e3) / 1e9,
                                  "wall_ms_incl_pack_decode": wall * 1e3, "aligned_by_dp": int((resd["whichFromDP"] < 2).sum())}
    dp.close()
    bb.close()
    return res


def cpu_baseline(hi, q, lens, wpq, mm, max_out, budget_s):
    """oracle (CPU restatement of the reference path) timed on a bounded sample with every host core"""
    import concurrent.futures as cf
    import s3o
    cores = os.cpu_count() or 1
    oix = s3o.Index(host_index=hi)
    n_all = len(lens)
    # calibrate on 512 reads, one thread
    t0 = time.perf_counter()
    oix.align_single_allbest(q, lens, wpq, mm, max_out, first=0, count=min(512, n_all))
    rate1 = min(512, n_all) / max(1e-6, time.perf_counter() - t0)
    per_thread = int(max(256, min(rate1 * budget_s, n_all // cores)) // 32 * 32)
    per_thread = max(32, per_thread)
    threads = max(1, min(cores, n_all // per_thread))

    def work(t):
        return oix.align_single_allbest(q, lens, wpq, mm, max_out, first=t * per_thread, count=per_thread)[6].rankQueries

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        ranks = list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    oix.close()
    total = per_thread * threads
    return {"value": total / dt / 1e6, "unit": "M reads/s", "cores": threads, "kind": "port",
            "sample": "%d reads (%d per thread x %d threads) of the same batch, same ladder, %.1f s wall; oracle/s3_oracle.c" %
                      (total, per_thread, threads, dt),
            "reads_per_s_per_core": total / dt / threads, "rank_queries_per_read": sum(ranks) / total}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-bp", type=int, default=int(os.environ.get("S3DP_BENCH_GENOME_BP", 3_100_000_000)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("S3DP_BENCH_READS", 10_000_000)), help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mismatch", type=int, default=4)
    ap.add_argument("--max-out", type=int, default=1000)
    ap.add_argument("--repeat-frac", type=float, default=0.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--check", type=int, default=2048, help="reads verified against the oracle after the run")
    ap.add_argument("--extras", type=int, default=1, help="after the timed region (N = 1): paired-end 2x150 BWT stage and DP kernel throughput")
    a = ap.parse_args()

    import torch
    import s3dp
    from s3dp import builder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or s3dp.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="nccl", device_id=dev)

    t_all = time.perf_counter()
    # ---- index (replicated per GPU): same seed on every rank -----------------------------------------
    t0 = time.perf_counter()
    codes = make_genome(a.genome_bp, 1, dev, a.repeat_frac)
    log("genome %d bp generated in %.1f s" % (a.genome_bp, time.perf_counter() - t0))
    t0 = time.perf_counter()
    reads2d, tpos, tstrand, tsub = make_reads(codes, a.reads, a.read_len, 2026 + rank, dev)
    wpq = s3dp.word_per_query(a.read_len)
    q = pack_on_device(reads2d, wpq)
    lens = np.full(a.reads, a.read_len, dtype=np.uint32)
    del reads2d
    log("%d reads generated + packed in %.1f s" % (a.reads, time.perf_counter() - t0))
    t0 = time.perf_counter()
    hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=(rank == 0))
    if not (rank == 0 and world == 1 and a.extras):
        del codes
    torch.cuda.empty_cache()
    log("index built on the GPU in %.1f s" % (time.perf_counter() - t0))
    t0 = time.perf_counter()
    ix = s3dp.Index(hi, device=local)
    log("index resident: %.2f GB in HBM, upload %.1f s" % (ix.device_bytes() / 1e9, time.perf_counter() - t0))
    b = s3dp.Batch(ix, a.reads, wpq)
    b.upload(q, lens)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- one counted pass: algorithmic work (N_rank etc.) ------------------------------------------------
    b.enable_counters(True)
    b.align_single_allbest(a.mismatch, a.max_out)
    b.sync()
    cnt = b.counters()
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    aligned = int((phase > 0).sum())
    n_occ = int(offs[-1])
    b.enable_counters(False)

    # ---- timed region ------------------------------------------------------------------------------------
    for _ in range(a.warmup):
        b.align_single_allbest(a.mismatch, a.max_out)
        b.sync()
    barrier()
    t0 = time.perf_counter()
    search_ms = 0.0
    ce_ms = 0.0
    dev_ms = 0.0
    launches = 0
    for _ in range(a.steps):
        b.align_single_allbest(a.mismatch, a.max_out)
        b.sync()
        tot_ms, s_ms, k = b.timing()        # HIP events on the batch stream
        search_ms += s_ms
        dev_ms += tot_ms
        ce_ms += b.ce_timing()[0]
        launches += k
    barrier()
    dt = time.perf_counter() - t0
    from s3dp import dist as sd
    dt = sd.max_over_ranks(dt, device=dev)
    # the only collective of the path: the final reduce of the counters (RCCL all-reduce of 5 numbers)
    tot = sd.reduce_counters({"reads": a.reads, "readsAligned": aligned, "alignments": n_occ, "rankQueries": cnt["rankQueries"],
                              "records": cnt["records"]}, device=dev)
    total_reads = tot["reads"]
    value = total_reads * a.steps / dt / 1e6

    # ---- roofline of the search (stepping kernel + check-and-extend kernel: the two halves of one search pass) ----------------
    # algorithmic bytes (extended by what check-and-extend reads since it became half of the pass): 32 B per rank query, 16 B per lookup
    # (one long-table entry, or the four 4-byte entries of the 13-mer tables), 4 B per suffix-array entry, 16 B per 32-character compare window
    # (8 B of packed text + 8 B of the read), packed read + length per read, 12 B per record
    bytes_step = 32.0 * cnt["rankQueries"] + 16.0 * cnt["ltLookups"] + 4.0 * cnt["saReads"] + 16.0 * cnt["ceWindows"] + \
        a.reads * (4.0 * wpq + 4.0) + 12.0 * cnt["records"]
    search_s_per_step = (search_ms + ce_ms) / 1e3 / max(1, a.steps)
    achieved = bytes_step / search_s_per_step / 1e9 if search_s_per_step > 0 else 0.0
    # HBM traffic of the two kernels comes from a separate rocprofv3 --pmc FETCH_SIZE pass (it cannot be collected from inside this process);
    # the committed measurement is attached when it was taken on exactly this workload, per launch like `achieved`
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for w in json.load(f)["workloads"]:
                if all(w[k] == v for k, v in (("genome_bp", a.genome_bp), ("reads_per_gpu", a.reads), ("read_len", a.read_len),
                                              ("max_mismatch", a.mismatch), ("max_output_per_read", a.max_out), ("repeat_frac", a.repeat_frac))):
                    traffic = w["fetch_bytes_per_step"] / w["launches_per_step"]
    except (OSError, ValueError, KeyError):
        traffic = None
    n_launch = max(1, launches // max(1, a.steps))
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": "bytes per launch pair (FETCH_SIZE of both kernels, profiles/pmc_traffic.json)",
            "algorithmic_bytes_per_launch": bytes_step / n_launch,
            "kernel": "s3::searchKernelV2 + s3::ceKernel", "launches_per_step": n_launch,
            "kernel_ms_per_step": (search_ms + ce_ms) / max(1, a.steps),
            "stepping_kernel_ms_per_step": search_ms / max(1, a.steps),
            "check_extend_kernel_ms_per_step": ce_ms / max(1, a.steps), "device_ms_per_step": dev_ms / max(1, a.steps),
            "sa_reads_per_step": cnt["saReads"], "ce_windows_per_step": cnt["ceWindows"], "lt_lookups_per_step": cnt["ltLookups"],
            "rank_queries_per_step": cnt["rankQueries"]}
    if traffic:
        # the path is made of random accesses with 4-32 B payloads; the GPU moves 64 B per access at ~50 G accesses/s (tools/hbm_probe.py), so the
        # rate of fetched 64-byte lines is what tells how busy the memory system is
        lines = traffic * n_launch / 64.0 / search_s_per_step / 1e9
        roof["random_line_rate"] = {"achieved_Glines_per_s": lines, "probe_Glines_per_s": 50.0, "frac": lines / 50.0}

    out = {"metric": "M reads/sec aligned (100 bp SE, GRCh38-scale synthetic genome)", "value": value, "unit": "M reads/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "synthetic uniform-ACGT genome, full-SA 2BWT+GPU-2BWT index in HBM; 100 bp single-end reads, "
                                  "0-4 substitutions (55/25/12/5/3 %), <=4 mismatch all-best ladder (-s 4 -h 2), DP off; "
                                  "BASELINE.json configs[1] with GRCh38 replaced by a synthetic genome of the same scale",
                      "genome_bp": a.genome_bp, "reads_per_gpu": a.reads, "read_len": a.read_len, "max_mismatch": a.mismatch,
                      "max_output_per_read": a.max_out, "repeat_frac": a.repeat_frac, "parallelism": "reads sharded x%d, index replicated" % world,
                      "lookup_chars": int(s3dp.lib().s3_index_lookup_chars(ix.h))},
           "roofline": roof,
           "aligned_reads": tot["readsAligned"], "occurrences": tot["alignments"],
           "rank_queries_per_read": cnt["rankQueries"] / a.reads}

    # ---- correctness spot check + CPU baseline (rank 0, N = 1) ------------------------------------------
    if rank == 0:
        if a.check > 0:
            import s3o
            oix = s3o.Index(host_index=hi)
            m = min(a.check, a.reads) // 32 * 32
            oph, owe, ooff, opos, ostr, omis, _ = oix.align_single_allbest(q, lens, wpq, a.mismatch, a.max_out, first=0, count=m)
            ok = (phase[:m] == oph).all() and (werr[:m] == owe).all() and (offs[:m + 1] == ooff).all() and \
                (pos[:int(ooff[-1])] == opos).all() and (strand[:int(ooff[-1])] == ostr).all() and (mis[:int(ooff[-1])] == omis).all()
            out["oracle_check"] = {"reads": m, "bit_exact": bool(ok)}
            oix.close()
            if not ok:
                raise SystemExit("bench.py: HIP result differs from the CPU oracle")
        if world == 1 and a.cpu_budget > 0:
            out["cpu_baseline"] = cpu_baseline(hi, q, lens, wpq, a.mismatch, a.max_out, a.cpu_budget)
        if world == 1 and a.extras:
            try:
                out["extras"] = extras(s3dp, ix, codes, dev, rank)
            except Exception as e:                                      # secondary figures must never cost the headline line
                out["extras"] = {"error": repr(e)}
        out["wall_s_total"] = time.perf_counter() - t_all
        print(json.dumps(out), flush=True)
    b.close()
    ix.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
