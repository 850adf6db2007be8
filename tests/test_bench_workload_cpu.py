"""bench.py's repeat-rich workload and its truth check, on the CPU: a small genome with the planted repeat families, reads simulated by bench.py,
aligned by the oracle (the checker), judged by bench.truth_check.  Pins the conventions the check relies on (position = leftmost text position
on the forward strand, strand 1 / 2) and that the check notices a wrong answer."""
import numpy as np
import torch

import bench
import s3dp
import s3o
from s3dp import builder


def test_truth_check_on_a_repeat_rich_genome():
    dev = torch.device("cpu")
    codes, fam = bench.make_genome(1_500_000, 1, dev, 0.5)
    assert fam["sine_alu_like"]["copies"] > 100
    L, n = 100, 2048
    reads2d, tpos, tstrand, tsub = bench.make_reads(codes, n, L, 7, dev)
    wpq = s3dp.word_per_query(L)
    q = bench.pack_on_device(reads2d, wpq)
    lens = np.full(n, L, dtype=np.uint32)
    hi = builder.build_host_index(codes.numpy(), device="cpu")
    oix = s3o.Index(host_index=hi)
    ph, we, off, pos, st, mis, _ = oix.align_single_allbest(q, lens, wpq, 4, 1000, first=0, count=n)
    oix.close()
    t = bench.truth_check(ph, off, pos, st, mis, tpos.numpy(), tstrand.numpy(), tsub.numpy().astype(np.int64), 1000)
    assert t["ok"], t
    assert t["true_place_reported"] > n * 0.9
    assert t["multi_mapped_reads"] > 0          # the planted families make reads that map to several places
    # a shifted answer must be noticed
    bad = pos.copy()
    bad[: len(bad) // 2] += 1
    t2 = bench.truth_check(ph, off, bad, st, mis, tpos.numpy(), tstrand.numpy(), tsub.numpy().astype(np.int64), 1000)
    assert not t2["ok"] and t2["true_place_missing"] > 0
    # the stratified oracle check (bench.py's oracle_check): with MaxOutputPerRead 5 this genome has reads in every stratum; the oracle's own lists pass,
    # a changed occurrence of a read whose list was cut is found
    oix = s3o.Index(host_index=hi)
    ph, we, off, pos, st, mis, _ = oix.align_single_allbest(q, lens, wpq, 4, 5, first=0, count=n)
    oix.close()
    c = bench.stratified_oracle_check(hi, q, lens, wpq, 4, 5, ph, we, off, pos, st, mis, 512, 60.0)
    assert c["bit_exact"] and c["reads"] >= 128, c
    assert c["strata"]["cut_at_max_output"]["checked"] >= 32 and c["strata"]["first_reads"]["checked"] >= 32, c
    counts = np.diff(off.astype(np.int64))
    cut = np.nonzero(counts >= 5)[0]
    bad = pos.copy()
    for r in cut:
        bad[int(off[r]) + 3] ^= 1
    c2 = bench.stratified_oracle_check(hi, q, lens, wpq, 4, 5, ph, we, off, bad, st, mis, 512, 60.0)
    assert not c2["bit_exact"] and set(c2["differing_reads"]) <= set(int(x) for x in cut)


def test_cpu_baseline_leg_runs_with_and_without_a_cpu_quota(monkeypatch):
    """bench.cpu_baseline (the "port" leg): a bounded sample on every core the process may use -- the hardware's, cut to the control group's quota"""
    import builtins
    import io
    dev = torch.device("cpu")
    codes, _ = bench.make_genome(400_000, 3, dev, 0.0)
    L, n = 100, 1024
    reads2d, _, _, _ = bench.make_reads(codes, n, L, 9, dev)
    wpq = s3dp.word_per_query(L)
    q = bench.pack_on_device(reads2d, wpq)
    lens = np.full(n, L, dtype=np.uint32)
    hi = builder.build_host_index(codes.numpy(), device="cpu")
    real_open = builtins.open
    for text, want in (("200000 100000\n", 2), ("max 100000\n", None), (None, None)):
        def fake(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                if text is None:
                    raise OSError("no such file")
                return io.StringIO(text)
            return real_open(path, *a, **k)
        monkeypatch.setattr(builtins, "open", fake)
        r = bench.cpu_baseline(hi, q, lens, wpq, 4, 1000, 2.0)
        monkeypatch.setattr(builtins, "open", real_open)
        assert r["value"] > 0 and r["rank_queries_per_read"] > 0
        if want:
            assert r["cores"] == want
