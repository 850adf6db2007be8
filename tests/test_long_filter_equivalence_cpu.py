"""The long-read filter's outcome does not depend on the order of events in the reference's loop.

validateAlignments (CPUfunctions.cpp:1129-1222) walks a read's seed hits one by one: a hit with fewer seed mismatches than the phase's floor is skipped, one
whose seed + extension mismatches exceed the limit is skipped; in the best modes a hit worse than the best so far is skipped, a better one EMPTIES what
was kept, and the walk stops once the smallest possible total has filled the hit limit; the list is cut at the hit limit afterwards.  The device shares a
list among the lanes of a wave (validateBigKernel, s3dp_ladder.hip), which needs the outcome as a function of the list alone: the qualifying hits -- best
modes: those whose total equals the list's minimum -- in list order, the first `max_hit` of them.  This test holds the two statements against each other
on random lists (the GPU parity tests hold the kernel against the oracle's loop)."""
import random


def serial_filter(hits, min_seed_mis, max_mis, only_best, max_hit):
    """the reference's loop; hits = [(seed_mis, total or None when the hit cannot be extended inside the text)]"""
    kept, pre = [], max_mis
    for i, (seed_mis, total) in enumerate(hits):
        if seed_mis < min_seed_mis:
            continue
        tot = max_mis + 1 + seed_mis if total is None else total
        if tot > max_mis:
            continue
        if only_best and tot > pre:
            continue
        if only_best and tot < pre:
            kept, pre = [], tot
        kept.append((i, tot))
        if pre == min_seed_mis and len(kept) >= max_hit:
            break
    return kept[:max_hit]


def list_filter(hits, min_seed_mis, max_mis, only_best, max_hit):
    """what the wave computes: totals (None = does not qualify), their minimum, an ordered compaction cut at the limit"""
    tot = [None if (s < min_seed_mis or t is None or t > max_mis) else t for s, t in hits]
    have = [t for t in tot if t is not None]
    if not have:
        return []
    best = min(have)
    return [(i, t) for i, t in enumerate(tot) if t is not None and (not only_best or t == best)][:max_hit]


def test_serial_walk_and_list_formulation_agree():
    rng = random.Random(20261003)
    cases = 0
    for _ in range(20000):
        n = rng.choice([0, 1, 2, 5, 9, 64, 65, 200])
        max_mis = rng.choice([1, 3, 4, 6])
        only_best = rng.random() < 0.5
        min_seed_mis = rng.randint(0, 4) if only_best else 0          # best modes: the phase's mismatch count; otherwise 0
        max_hit = rng.choice([1, 2, 5, 40, 1000])
        hits = []
        for _ in range(n):
            seed = rng.randint(0, 4)
            ext = rng.choice([0, 0, 0, 1, 2, 5])
            hits.append((seed, None if rng.random() < 0.05 else seed + ext))
        a = serial_filter(hits, min_seed_mis, max_mis, only_best, max_hit)
        b = list_filter(hits, min_seed_mis, max_mis, only_best, max_hit)
        assert a == b, (hits, min_seed_mis, max_mis, only_best, max_hit, a, b)
        cases += bool(a)
    assert cases > 5000
