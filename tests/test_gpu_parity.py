"""-m gpu: the HIP path, called through the C-ABI, against the CPU oracle and the reference's golden dumps."""
import os

import numpy as np
import pytest

import s3dp
from s3dp import builder
import s3o
from conftest import GOLDEN, golden_reads, hits_to_lines, host_index, read_gz_lines

pytestmark = pytest.mark.gpu

_DEV = {}


def dev_index(name):
    if name not in _DEV:
        _DEV[name] = s3dp.Index(host_index(name))
    return _DEV[name]


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120), ("g6", 100)])     # g6: the ends of the text, padded lookup keys (tools/simulate_edge_case.py)
@pytest.mark.parametrize("mm", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("heavy", ["0", "1", "2", "2i", "2g", "2p"])
def test_search_bit_exact_vs_reference_dump_and_oracle(monkeypatch, name, maxlen, mm, heavy):
    """heavy: S3DP_HEAVY -- 0: every read is walked by one lane; 1 (the default): a read that reserves more than 256 check-and-extend candidates
    leaves the first pass and is taken apart by a second launch of the kernel, one lane per (item, position of the first mismatch) unit; 2: every
    read with more than two candidates goes that way; 2i: the same with whole items as units (S3DP_UNITS=0).  The records must not depend on it."""
    monkeypatch.setenv("S3DP_HEAVY", heavy[0])
    monkeypatch.setenv("S3DP_UNITS", "0" if heavy.endswith("i") else "1")
    if heavy.endswith("g"):                                 # no flank tables: every candidate of ceKernel goes through the suffix array and the text
        monkeypatch.setenv("S3DP_FLANK", "0")
    if heavy.endswith("p"):                                 # one first-mismatch position per unit (until round 5; the default: groups of two for the single-end ladders, one position for pairs)
        monkeypatch.setenv("S3DP_UNIT_GROUP", "1")
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.enable_counters(True)
    b.search(mm, 1000)
    hits, offs, tot, werr = b.fetch_hits()
    mine = hits_to_lines(hits, offs, tot, werr, len(reads))
    ref = read_gz_lines(os.path.join(GOLDEN, name, "search_mm%d.txt.gz" % mm))
    assert mine == ref
    # full record order (including interleaving of SA ranges and check-and-extend hits) vs the oracle
    oix = s3o.Index(host_index=host_index(name))
    ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, mm, max_out=1000)
    assert (offs == ooffs).all()
    assert hits.tobytes() == ohits.tobytes()
    assert (tot == otot).all() and (werr == owerr).all()
    c = b.counters()
    if heavy == "0":                                        # (a read that takes the exact pass is searched twice: the counts are for the single pass)
        # single-suffix ranges leave the exact lookup step early (same records, fewer rank queries); S3DP_EARLY_CE=0 walks like the CPU path
        if os.environ.get("S3DP_EARLY_CE") == "0":
            assert c["rankQueries"] == st.rankQueries
        else:
            assert c["rankQueries"] <= st.rankQueries
        if os.environ.get("S3DP_LT_EXTRA", "0") == "0":
            assert c["ltLookups"] == st.ltLookups               # the long tables answer an item with one entry
        else:
            assert c["ltLookups"] <= st.ltLookups
    oix.close()
    b.close()


KNOBS = [
    {"S3DP_EARLY_CE": "0"},                                  # the exact lookup step is walked to its end, as the CPU path walks it
    {"S3DP_EARLY_CE": "4", "S3DP_EARLY_MIN": "17"},
    {"S3DP_HEAVY": "1", "S3DP_HEAVY_RANK": "8"},             # nearly every read is listed by its rank queries
    {"S3DP_HEAVY": "2", "S3DP_FLANK": "1"},                  # flank table of the forward index only: tasks of forward steps take the suffix array and the text
    {"S3DP_HEAVY": "0", "S3DP_FLANK": "0", "S3DP_EARLY_CE": "0"},
    {"S3DP_HEAVY": "2", "S3DP_UNIT_NODES": "0"},             # no adoption of frames: every unit is searched by one lane
    {"S3DP_HEAVY": "2", "S3DP_UNIT_NODES": "300"},           # the counters of adopted frames run out on the way
    {"S3DP_HEAVY": "2", "S3DP_BLOCKS_PER_CU": "1", "S3DP_CE_BLOCKS_PER_CU": "1"},
    # round 5: every trip that leaves a stepping lane is followed by trips that only step (the fixtures never gather the 32 stepping lanes of the default)
    {"S3DP_HEAVY": "2", "S3DP_FAST_MIN": "1", "S3DP_FAST_TRIPS": "3", "S3DP_FIRST_FAST_MIN": "1", "S3DP_FIRST_FAST_TRIPS": "3"},
    {"S3DP_HEAVY": "1", "S3DP_HEAVY_RANK": "8", "S3DP_FAST_MIN": "2", "S3DP_FAST_TRIPS": "15", "S3DP_FIRST_FAST_MIN": "2", "S3DP_FIRST_FAST_TRIPS": "15"},
    {"S3DP_HEAVY": "2", "S3DP_FAST_TRIPS": "0", "S3DP_FIRST_FAST_TRIPS": "0"},     # none at all (rounds 3-4)
    # every lane with a frame to give is a giant after its first full trip: frames are adopted while the queue still has work
    {"S3DP_HEAVY": "2", "S3DP_GIANT_AGE": "1", "S3DP_GIANT_IDLE": "1", "S3DP_FAST_MIN": "4"},
    {"S3DP_HEAVY": "2", "S3DP_GIANT_AGE": "0", "S3DP_TAIL_AGE": "2"},
    # units are groups of consecutive first-mismatch positions (default 2 for the single-end ladders, 1 for pairs): one position per unit (until round 5), odd sizes, and one group for the whole step
    {"S3DP_HEAVY": "2", "S3DP_UNIT_GROUP": "1"},
    {"S3DP_HEAVY": "2", "S3DP_UNIT_GROUP": "3", "S3DP_GIANT_AGE": "1", "S3DP_GIANT_IDLE": "1", "S3DP_FAST_MIN": "2"},
    {"S3DP_HEAVY": "1", "S3DP_HEAVY_RANK": "8", "S3DP_UNIT_GROUP": "7"},
    {"S3DP_HEAVY": "2", "S3DP_UNIT_GROUP": "200"},
    {"S3DP_HEAVY": "2", "S3DP_UNIT_GROUP_HI": "2"},           # the same group size whatever the mismatch budget (until round 6; default: 8 from three mismatches on)
    {"S3DP_HEAVY": "1", "S3DP_HEAVY_RANK": "8", "S3DP_UNIT_GROUP_HI": "5", "S3DP_FLANK": "1"},
]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join("%s=%s" % (a[5:], b) for a, b in k.items()))
@pytest.mark.parametrize("name,maxlen,mm", [("g2", 120, 2), ("g2", 120, 4), ("g1", 100, 3)])
def test_search_knobs_do_not_change_results(monkeypatch, name, maxlen, mm, knobs):
    """Every tuning knob of the search (thresholds of the heavy-read machinery, early check-and-extend, grid widths, the unit-pass kernel and its
    adoption of frames) at an extreme value: records, order and tallies stay the oracle's.  With S3DP_EARLY_CE=0 and one lane per read the kernels'
    rank-query count EQUALS the oracle's."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    if "S3DP_HEAVY" not in knobs:
        monkeypatch.setenv("S3DP_HEAVY", "0")
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.enable_counters(True)
    b.search(mm, 1000)
    hits, offs, tot, werr = b.fetch_hits()
    oix = s3o.Index(host_index=host_index(name))
    ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, mm, max_out=1000)
    assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes()
    assert (tot == otot).all() and (werr == owerr).all()
    c = b.counters()
    if knobs.get("S3DP_EARLY_CE") == "0":
        assert c["rankQueries"] == st.rankQueries
    b.enable_counters(False)
    b.search(mm, 7)                                          # and with a tight MaxOutputPerRead, uncounted kernels
    hits, offs, tot, werr = b.fetch_hits()
    ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, mm, max_out=7)
    assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes()
    assert (tot == otot).all() and (werr == owerr).all()
    oix.close()
    b.close()


def test_search_grid_width_does_not_change_results(monkeypatch):
    """s3_batch_set_grid: the paired command line narrows its search grids to leave room for the DP kernels; one workgroup per compute unit, three, and all again"""
    monkeypatch.setenv("S3DP_HEAVY", "2")
    reads = golden_reads("g2")
    q, lens, wpq = s3dp.pack_reads(reads, 120)
    b = s3dp.Batch(dev_index("g2"), len(reads), wpq)
    b.upload(q, lens)
    oix = s3o.Index(host_index=host_index("g2"))
    ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, 3, max_out=1000)
    for width in (1, 3, 0):
        b.set_grid(width)
        b.search(3, 1000)
        hits, offs, tot, werr = b.fetch_hits()
        assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes()
        assert (tot == otot).all() and (werr == owerr).all()
    with pytest.raises(s3dp.S3Error):
        b.set_grid(17)
    oix.close()
    b.close()


@pytest.mark.parametrize("extra", [1, 2, 3])
@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
def test_long_lookup_tables_do_not_change_results(monkeypatch, name, maxlen, extra):
    """S3DP_LT_EXTRA: 14- / 15- / 16-mer tables built in HBM at upload (the state after the 13-mer lookup plus one / two / three exact steps).  Records, their
    order and the per-read tallies must stay bit-identical to the oracle for every mismatch budget; only the number of rank queries drops."""
    monkeypatch.setenv("S3DP_HEAVY", "0")                   # the rank-query comparison below is for a single pass over every read
    monkeypatch.setenv("S3DP_LT_EXTRA", str(extra))
    ix = s3dp.Index(host_index(name))
    assert s3dp.lib().s3_index_lookup_chars(ix.h) == 13 + extra
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    b.enable_counters(True)
    oix = s3o.Index(host_index=host_index(name))
    for mm in range(5):
        b.search(mm, 1000)
        hits, offs, tot, werr = b.fetch_hits()
        ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, mm, max_out=1000)
        assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes()
        assert (tot == otot).all() and (werr == owerr).all()
        c = b.counters()
        assert c["rankQueries"] <= st.rankQueries and (mm == 0 or c["rankQueries"] < st.rankQueries)
        assert c["ltLookups"] <= st.ltLookups
    b.align_single_allbest(4, 1000)
    phase, we, off2, pos, strand, mis = b.fetch_alignments()
    oph, owe, ooff, opos, ostr, omis, _ = oix.align_single_allbest(q, lens, wpq, 4, 1000)
    assert (phase == oph).all() and (off2 == ooff).all() and (pos[:int(ooff[-1])] == opos).all() and (mis[:int(ooff[-1])] == omis).all()
    oix.close()
    b.close()
    ix.close()


@pytest.mark.parametrize("cap", [1, 3, 17])
@pytest.mark.parametrize("heavy", ["0", "1", "3", "3i", "3g"])
def test_max_output_truncation(monkeypatch, cap, heavy):
    monkeypatch.setenv("S3DP_HEAVY", heavy[0])
    monkeypatch.setenv("S3DP_UNITS", "0" if heavy.endswith("i") else "1")
    if heavy.endswith("g"):
        monkeypatch.setenv("S3DP_FLANK", "0")
    name, maxlen, mm = "g2", 120, 2
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.search(mm, cap)
    hits, offs, tot, werr = b.fetch_hits()
    oix = s3o.Index(host_index=host_index(name))
    ohits, ooffs, otot, owerr, st = oix.search(q, lens, wpq, mm, max_out=cap)
    assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes()
    assert (tot == otot).all() and (werr == owerr).all()
    assert tot.max() <= cap
    oix.close()
    b.close()


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
@pytest.mark.parametrize("mm", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("heavy", ["1", "2", "2m", "2f"])
def test_allbest_ladder_matches_oracle(monkeypatch, name, maxlen, mm, heavy):
    """heavy 2: nearly every read is listed for the second pass -- and, listed in one phase, listed at once in the next (the ladder's memory of its heavy
    reads, round 5); 2m: without that memory; 2f: with fast trips behind every trip that leaves a stepping lane"""
    monkeypatch.setenv("S3DP_HEAVY", heavy[0])
    if heavy.endswith("m"):
        monkeypatch.setenv("S3DP_HEAVY_MEMO", "0")
    if heavy.endswith("f"):
        for k in ("S3DP_FAST_MIN", "S3DP_FIRST_FAST_MIN"):
            monkeypatch.setenv(k, "1")
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.align_single_allbest(mm, 1000)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    oix = s3o.Index(host_index=host_index(name))
    ophase, owerr, ooffs, opos, ostrand, omis, st = oix.align_single_allbest(q, lens, wpq, mm, 1000)
    assert (phase == ophase).all()
    assert (werr == owerr).all()
    assert (offs == ooffs).all()
    assert (pos == opos).all() and (strand == ostrand).all() and (mis == omis).all()
    oix.close()
    b.close()


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
@pytest.mark.parametrize("report_type,mm", [(1, 0), (1, 2), (1, 4), (3, 0), (3, 3), (4, 2), (4, 4)])
def test_report_type_ladders_match_oracle(name, maxlen, report_type, mm):
    """-h 1 (all valid: one terminal phase), -h 3/4 (unique/random best: phases 0..k, phase reported + 1)"""
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.align_single(mm, 1000, report_type)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    oix = s3o.Index(host_index=host_index(name))
    ophase, owerr, ooffs, opos, ostrand, omis, st = oix.align_single_allbest(q, lens, wpq, mm, 1000, report_type=report_type)
    assert (phase == ophase).all()
    assert (werr == owerr).all()
    assert (offs == ooffs).all()
    assert (pos == opos).all() and (strand == ostrand).all() and (mis == omis).all()
    if report_type >= 3:
        n = offs[1:] - offs[:-1]
        assert ((phase > 0) == (n > 0)).all()
        for r in range(len(reads)):                       # every reported hit of a best mode carries exactly phase-1 mismatches
            assert (mis[offs[r]:offs[r + 1]].astype(int) == int(phase[r]) - 1).all()
    oix.close()
    b.close()


@pytest.mark.parametrize("report_type,mm", [(2, 3), (2, 1), (1, 2), (3, 2), (4, 3)])
def test_long_read_mode_matches_oracle(report_type, mm):
    """reads of 121..200 bases (-L 200): the first 100 bases are searched without a hit limit, the rest is verified on the device"""
    name, maxlen = "g3", 200
    reads = golden_reads(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    assert wpq == 16
    seeds = np.where(lens > 120, 100, lens).astype(np.uint32)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.set_seed_lengths(seeds)
    b.align_single(mm, 1000, report_type)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    oix = s3o.Index(host_index=host_index(name))
    ophase, owerr, ooffs, opos, ostrand, omis, st = oix.align_single_allbest(q, lens, wpq, mm, 1000, report_type=report_type, seed_lengths=seeds)
    assert (phase == ophase).all()
    assert (offs == ooffs).all()
    assert (pos == opos).all() and (strand == ostrand).all() and (mis == omis).all()
    assert (offs[1:] - offs[:-1]).max() > 1 and phase.max() > 0
    # the same batch without seeds must still behave as a plain batch afterwards
    b.set_seed_lengths(None)
    oix.close()
    b.close()


@pytest.mark.parametrize("report_type,mm,max_out,budget", [(2, 2, 1000, 0), (1, 2, 1000, 0), (2, 3, 40, 0), (3, 2, 1000, 0), (4, 2, 5, 0),
                                                          (1, 2, 1000, 700), (2, 3, 40, 5000), (3, 2, 1000, 1), (4, 2, 5, 20000)])
def test_long_read_mode_inside_repeat_families(monkeypatch, report_type, mm, max_out, budget):
    """150-base reads (-L 150: the first 100 bases searched without a hit limit, the other 50 verified on the device) from a dispersed family of 600
    near-identical copies and a tandem array: the seed lists hold hundreds of hits, which the verification (validateKernel, s3dp_ladder.hip) shares
    among the lanes of a wave; lists of up to eight hits stay with one lane.  What is kept, its order, the shift of reverse hits to the read's leftmost
    base, the cut at the hit limit and the best modes' "fewest mismatches only" must equal the reference's serial filter (validateAlignments,
    CPUfunctions.cpp:1129-1222) -- at a hit limit above and below the list lengths, and with reads hanging over both ends of the text."""
    if budget:
        # round 5: a pass that decodes more seed hits than S3DP_LONG_BUDGET works them off in slices of reads (scratch lists, only the kept hits stay): at
        # 700 / 5000 / 20000 hits per slice, and with every read a slice of its own, nothing may change
        monkeypatch.setenv("S3DP_LONG_BUDGET", str(budget))
    rng = np.random.default_rng(78)
    n, L = 300_000, 150
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    unit = rng.integers(0, 4, size=700, dtype=np.uint8)
    for c in range(550):                                      # dispersed family: the seeds' side 0.3 % from the consensus, the far side 3 %
        at = 1000 + c * 500 + int(rng.integers(0, 40))
        cp = unit.copy()
        rate = np.where(np.arange(700) % 300 < 150, 0.003, 0.03)
        flip = rng.random(700) < rate
        cp[flip] = (cp[flip] + rng.integers(1, 4, size=int(flip.sum()))) % 4
        codes[at:at + 450] = cp[:450]
    sat = rng.integers(0, 4, size=171, dtype=np.uint8)
    for c in range(100):                                      # tandem array at the very end of the text, 1 % divergence
        cp = sat.copy()
        flip = rng.random(171) < 0.01
        cp[flip] = (cp[flip] + rng.integers(1, 4, size=int(flip.sum()))) % 4
        codes[n - (c + 1) * 171:n - c * 171] = cp
    codes[:171 * 3] = np.tile(sat, 3)                          # and three units at the very start
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    reads = []
    for k in range(384):
        where = k % 4
        if where == 0:
            start = 1000 + int(rng.integers(0, 550)) * 500 + int(rng.integers(0, 250))
        elif where == 1:
            start = n - 171 * 100 + int(rng.integers(0, 171 * 100 - L))
        elif where == 2:
            start = int(rng.integers(0, n - L))
        else:
            start = int(rng.integers(0, 171 * 3 - L)) if k % 8 == 3 else n - L - int(rng.integers(0, 171))
        rd = codes[start:start + L].copy()
        for _ in range(int(rng.integers(0, 4))):
            j = int(rng.integers(0, L))
            rd[j] = (rd[j] + 1 + int(rng.integers(0, 3))) % 4
        if k % 3 == 0:
            rd = comp[rd[::-1]]
        reads.append(bytes(b"ACGT"[c] for c in rd))
    q, lens, wpq = s3dp.pack_reads(reads, L)
    seeds = np.where(lens > 120, 100, lens).astype(np.uint32)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    b.set_seed_lengths(seeds)
    b.align_single(mm, max_out, report_type)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    oix = s3o.Index(host_index=hi)
    ophase, owerr, ooffs, opos, ostrand, omis, st = oix.align_single_allbest(q, lens, wpq, mm, max_out, report_type=report_type, seed_lengths=seeds)
    nhit = ooffs[1:] - ooffs[:-1]
    if max_out >= 1000 and report_type <= 2:
        assert (nhit > 64).sum() > 20, "the workload should hold reads with lists longer than a wave"
    assert (nhit > 0).sum() > 200 and ((nhit > 0) & (nhit <= 8)).sum() > 50
    assert (phase == ophase).all()
    bad = np.nonzero((offs[1:] - offs[:-1]) != nhit)[0]
    assert len(bad) == 0, "read %d keeps %d hits, the oracle %d" % (bad[0], (offs[1:] - offs[:-1])[bad[0]], nhit[bad[0]])
    assert (pos == opos).all() and (strand == ostrand).all() and (mis == omis).all()
    oix.close()
    b.close()
    ix.close()


def test_buffers_grow_instead_of_overflowing(monkeypatch):
    """record and occurrence buffers start tiny (env) and must grow transparently: same result as the oracle"""
    monkeypatch.setenv("S3DP_REC_PER_READ", "0")
    monkeypatch.setenv("S3DP_REC_FLOOR", "0")
    monkeypatch.setenv("S3DP_OCC_PER_READ", "0")
    name, maxlen, mm, copies = "g2", 120, 3, 8
    one = golden_reads(name)
    n1 = len(one)
    assert n1 % 32 != 0                                   # copies do not line up with the 32-read interleave: pack them all
    q, lens, wpq = s3dp.pack_reads(one * copies, maxlen)
    b = s3dp.Batch(dev_index(name), n1 * copies, wpq)
    b.upload(q, lens)
    b.align_single(mm, 1000, 2)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    q1, lens1, _ = s3dp.pack_reads(one, maxlen)
    oix = s3o.Index(host_index=host_index(name))
    ophase, owerr, ooffs, opos, ostrand, omis, st = oix.align_single_allbest(q1, lens1, wpq, mm, 1000)
    assert (phase.reshape(copies, n1) == ophase).all()
    assert ((offs[1:] - offs[:-1]).reshape(copies, n1) == (ooffs[1:] - ooffs[:-1])).all()
    assert int(offs[-1]) == copies * len(opos) and int(offs[-1]) > 65536 * 2
    for c in (0, copies - 1):
        lo, hi = int(offs[c * n1]), int(offs[(c + 1) * n1])
        assert (pos[lo:hi] == opos).all() and (strand[lo:hi] == ostrand).all() and (mis[lo:hi] == omis).all()
    oix.close()
    b.close()


def golden_pairs(name):
    a = s3o.read_fastq(os.path.join(GOLDEN, name, "reads_1.fq"))
    b = s3o.read_fastq(os.path.join(GOLDEN, name, "reads_2.fq"))
    return [x for ab in zip(a, b) for x in ab]


@pytest.mark.parametrize("report_type,mm", [(2, 3), (2, 1), (2, 4), (2, 0), (1, 2), (3, 3), (4, 2), (4, 0)])
def test_pair_ladder_matches_oracle(report_type, mm):
    """paired-end ladder with the DP stages off: per-pair decisions, pairing, optimal pair, XA pairs, improper-pair lists"""
    name, maxlen = "g4", 100
    reads = golden_pairs(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    b.align_pair(mm, 8000, report_type, 200, 500, 1000)
    got = b.fetch_pairs()
    oix = s3o.Index(host_index=host_index(name))
    want = oix.align_pair(q, lens, wpq, mm, 8000, report_type, 200, 500, 1000)
    gp, wp = got[0], want[0]
    for f in gp.dtype.names:
        if f == "pad":
            continue
        bad = np.nonzero(gp[f] != wp[f])[0]
        assert len(bad) == 0, "pair field %s differs at pair %d: %s vs %s" % (f, bad[0], gp[bad[0]], wp[bad[0]])
    for k in range(1, 7):
        assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all(), "output array %d differs" % k
    assert (gp["phase"] > 0).sum() > 100 and (gp["phase"] == 0).sum() > 50
    oix.close()
    b.close()


def test_pair_small_caps():
    """tiny per-end hit limit and XA limit: the cuts must fall where the reference's fall"""
    name, maxlen = "g4", 100
    reads = golden_pairs(name)
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(dev_index(name), len(reads), wpq)
    b.upload(q, lens)
    oix = s3o.Index(host_index=host_index(name))
    for max_hits, pe_max in ((3, 2), (1, 1), (50, 3)):
        b.align_pair(2, max_hits, 2, 200, 500, pe_max)
        got = b.fetch_pairs()
        want = oix.align_pair(q, lens, wpq, 2, max_hits, 2, 200, 500, pe_max)
        for f in got[0].dtype.names:
            if f != "pad":
                assert (got[0][f] == want[0][f]).all(), (max_hits, pe_max, f)
        for k in range(1, 7):
            assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all(), (max_hits, pe_max, k)
    oix.close()
    b.close()


def test_pairs_inside_repeat_families_long_lists():
    """Pairs whose ends lie in a dispersed repeat family (600 near-identical copies) or in a tandem array: hundreds to thousands of hits per end, thousands
    of valid pairings.  Such pairs are taken by one wave each (pairStatsHeavyKernel / pairEmitHeavyKernel, s3dp_pair.hip) instead of one lane's serial
    sweep: the optimal pair, its index, the statistics, the "suboptimal" sum and the XA list in pairing order must equal the oracle's sweep
    (PEMappingCore / PEStatsPEPairList, PEAlgnmt.cpp:201-291,777-832) -- with every pair on the wave path (S3DP_PAIR_HEAVY=0), with none (4294967295)
    and with the default threshold."""
    rng = np.random.default_rng(77)
    n, L = 400_000, 100
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    unit = rng.integers(0, 4, size=700, dtype=np.uint8)
    for c in range(600):                                      # dispersed family: copies 0.3 % from the consensus
        at = 1000 + c * 500 + int(rng.integers(0, 40))
        cp = unit.copy()
        flip = rng.random(700) < 0.003
        cp[flip] = (cp[flip] + rng.integers(1, 4, size=int(flip.sum()))) % 4
        codes[at:at + 450] = cp[:450] if c % 3 else cp[250:700]
    sat = rng.integers(0, 4, size=171, dtype=np.uint8)
    for c in range(400):                                      # tandem array, 1 % divergence
        cp = sat.copy()
        flip = rng.random(171) < 0.01
        cp[flip] = (cp[flip] + rng.integers(1, 4, size=int(flip.sum()))) % 4
        codes[320_000 + c * 171:320_000 + (c + 1) * 171] = cp
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    reads = []
    for k in range(256):
        where = k % 4
        if where == 0:
            start = 1000 + int(rng.integers(0, 600)) * 500 + int(rng.integers(0, 60))
        elif where == 1:
            start = 320_000 + int(rng.integers(0, 171 * 396))
        elif where == 2:
            start = int(rng.integers(0, n - 600))
        else:
            start = 1000 + int(rng.integers(0, 590)) * 500 + 200
        ins = int(rng.integers(210, 420))
        a = codes[start:start + L].copy()
        bb = comp[codes[start + ins - L:start + ins][::-1]]
        for rd in (a, bb):
            for _ in range(int(rng.integers(0, 3))):
                j = int(rng.integers(0, L))
                rd[j] = (rd[j] + 1 + int(rng.integers(0, 3))) % 4
        pair = [bytes(b"ACGT"[c] for c in a), bytes(b"ACGT"[c] for c in bb)]
        if k % 5 == 0:
            pair.reverse()
        reads += pair
    q, lens, wpq = s3dp.pack_reads(reads, L)
    oix = s3o.Index(host_index=hi)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    keep = os.environ.get("S3DP_PAIR_HEAVY")
    try:
        for report_type, mm, pe_max in ((2, 2, 1000), (1, 2, 7), (4, 1, 1000)):
            want = oix.align_pair(q, lens, wpq, mm, 8000, report_type, 150, 500, pe_max)
            assert (want[0]["numPairs"] > 200).sum() > 10, "the workload should hold pairs with hundreds of pairings"
            for heavy in ("0", "4294967295", None):
                if heavy is None:
                    os.environ.pop("S3DP_PAIR_HEAVY", None)
                else:
                    os.environ["S3DP_PAIR_HEAVY"] = heavy
                b.align_pair(mm, 8000, report_type, 150, 500, pe_max)
                got = b.fetch_pairs()
                for f in got[0].dtype.names:
                    if f == "pad":
                        continue
                    bad = np.nonzero(got[0][f] != want[0][f])[0]
                    assert len(bad) == 0, "S3DP_PAIR_HEAVY=%s -h %d: pair field %s differs at pair %d: %s vs %s" % (heavy, report_type, f, bad[0], got[0][bad[0]], want[0][bad[0]])
                for k in range(1, 7):
                    assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all(), "S3DP_PAIR_HEAVY=%s -h %d: output array %d differs" % (heavy, report_type, k)
    finally:
        if keep is None:
            os.environ.pop("S3DP_PAIR_HEAVY", None)
        else:
            os.environ["S3DP_PAIR_HEAVY"] = keep
    oix.close()
    b.close()
    ix.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_text_ends_and_padded_lookup_keys_stress(seed):
    """A tiny genome searched with reads that hug or overhang its ends with runs of A -- the 13-mers the lookup tables count for the last / first text
    positions (padded with A), the in-text rule of the early check-and-extend hand-over, wrap-around of positions before the text start -- and with reads
    from its repeats.  Every record, order included, must equal the oracle's for every mismatch budget and for the all-best ladder."""
    rng = np.random.default_rng(seed)
    n, L = 4000, 100
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    codes[700:900] = codes[100:300]                          # a repeat
    codes[n - 9:] = 0                                        # the text ends with A's: padded and real 13-mers coincide
    codes[:5] = 0
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    reads = []
    for k in range(600):
        kind = k % 6
        if kind == 0:                                         # inside, near the end
            st = int(rng.integers(n - 140, n - L + 1)); seg = codes[st:st + L].copy()
        elif kind == 1:                                       # overhangs the end: the rest is A
            st = int(rng.integers(n - 95, n - 5)); seg = np.concatenate([codes[st:], np.zeros(L - (n - st), np.uint8)])
        elif kind == 2:                                       # overhangs the start
            ov = int(rng.integers(1, 60)); seg = np.concatenate([np.zeros(ov, np.uint8), codes[:L - ov]])
        elif kind == 3:                                       # near the start
            st = int(rng.integers(0, 40)); seg = codes[st:st + L].copy()
        elif kind == 4:                                       # from the repeat
            st = int(rng.integers(100, 200)); seg = codes[st:st + L].copy()
        else:
            st = int(rng.integers(0, n - L)); seg = codes[st:st + L].copy()
        for q in rng.choice(L, size=int(rng.integers(0, 4)), replace=False):
            seg[q] = (seg[q] + 1 + rng.integers(0, 3)) % 4
        if rng.integers(0, 2):
            seg = comp[seg[::-1]]
        reads.append(bytes(b"ACGT"[c] for c in seg))
    q, lens, wpq = s3dp.pack_reads(reads, L)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    oix = s3o.Index(host_index=hi)
    for mm in range(5):
        b.search(mm, 1000)
        hits, offs, tot, werr = b.fetch_hits()
        ohits, ooffs, otot, owerr, st_ = oix.search(q, lens, wpq, mm, max_out=1000)
        assert (offs == ooffs).all() and hits.tobytes() == ohits.tobytes(), mm
        assert (tot == otot).all() and (werr == owerr).all()
    b.align_single_allbest(4, 1000)
    phase, we, off2, pos, strand, mis = b.fetch_alignments()
    oph, owe, ooff, opos, ostr, omis, _ = oix.align_single_allbest(q, lens, wpq, 4, 1000)
    assert (phase == oph).all() and (off2 == ooff).all() and (pos[:int(ooff[-1])] == opos).all() and (mis[:int(ooff[-1])] == omis).all()
    oix.close(); b.close(); ix.close()
