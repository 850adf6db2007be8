"""random semi-global DP problems (reads cut from a window with substitutions / indels / clipped ends) in the reference packing"""
import numpy as np

import s3dp


def make_problems(n, max_read, max_dna, seed, read_len_range=(30, 150), clip=True, anchors=True, lengths=None, clip_max=None):
    """lengths: read lengths to cycle through instead of random ones; clip_max: soft-clip allowances up to min(clip_max, L // 2) instead of < 20"""
    rng = np.random.default_rng(seed)
    dna, reads, clip_lt, clip_rt, a_lt, a_rt, cut = [], [], [], [], [], [], []
    for t in range(n):
        L = int(rng.integers(read_len_range[0], min(read_len_range[1], max_read) + 1))
        if lengths is not None:
            L = int(lengths[t % len(lengths)])
        W = int(rng.integers(L + 2, max_dna + 1)) if max_dna > L + 2 else max_dna
        win = rng.integers(0, 4, size=W, dtype=np.uint8)
        kind = t % 8
        st = int(rng.integers(0, max(1, W - L)))
        read = win[st:st + L].copy()
        if len(read) < L:
            read = np.concatenate([read, rng.integers(0, 4, size=L - len(read), dtype=np.uint8)])
        if kind in (1, 5):                      # substitutions
            for p in rng.integers(0, L, size=int(rng.integers(1, 6))):
                read[p] = (read[p] + 1 + rng.integers(0, 3)) % 4
        if kind in (2, 5):                      # deletion from the read (reference has extra bases)
            d = int(rng.integers(1, 9)); p = int(rng.integers(5, L - 10))
            read = np.concatenate([read[:p], read[p + d:], rng.integers(0, 4, size=d, dtype=np.uint8)])[:L]
        if kind in (3, 6):                      # insertion into the read
            d = int(rng.integers(1, 9)); p = int(rng.integers(5, L - 10))
            read = np.concatenate([read[:p], rng.integers(0, 4, size=d, dtype=np.uint8), read[p:]])[:L]
        if kind in (4, 6):                      # junk ends -> soft clips
            e = int(rng.integers(2, 12))
            read[:e] = rng.integers(0, 4, size=e, dtype=np.uint8)
            read[L - e:] = rng.integers(0, 4, size=e, dtype=np.uint8)
        if kind == 7:                           # unrelated read
            read = rng.integers(0, 4, size=L, dtype=np.uint8)
        dna.append(win); reads.append(read)
        cmax = min(20, L // 3 + 1) if clip_max is None else max(1, min(clip_max, L // 2))
        clip_lt.append(int(rng.integers(0, cmax)) if clip else 0)
        clip_rt.append(int(rng.integers(0, cmax)) if clip else 0)
        if anchors and t % 3 == 0:
            a_lt.append(int(rng.integers(1, W + 1))); a_rt.append(int(rng.integers(0, W // 2 + 1)))
        else:
            a_lt.append(max_dna); a_rt.append(0)
        cut.append(int(np.ceil(0.3 * L)))
    return dict(dna=s3dp.pack_dp_sequences(dna, max_dna), dna_len=np.array([len(x) for x in dna], dtype=np.uint32),
                reads=s3dp.pack_dp_sequences(reads, max_read), read_len=np.array([len(x) for x in reads], dtype=np.uint32),
                clip_lt=np.array(clip_lt, dtype=np.uint32), clip_rt=np.array(clip_rt, dtype=np.uint32),
                a_lt=np.array(a_lt, dtype=np.uint32), a_rt=np.array(a_rt, dtype=np.uint32), cutoffs=np.array(cut, dtype=np.int32),
                raw_dna=dna, raw_reads=reads)
