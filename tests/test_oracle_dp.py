"""DP oracle (oracle/s3_oracle_dp.c) on the CPU: self-consistency (the score recomputed from the emitted pattern equals the reported
score) and the known answers quoted in SURVEY.md 8c-5.  The bit-for-bit check against the reference's own DP device code needs a GPU
(the reference has no CPU implementation): tests/test_gpu_dp_reference.py."""
import numpy as np

import s3dp
import s3o
from dp_cases import make_problems

SC = (1, -2, -3, -1)


def run(dna_list, read_list, max_dna, max_read, scheme=1, **kw):
    dna = s3dp.pack_dp_sequences(dna_list, max_dna)
    rd = s3dp.pack_dp_sequences(read_list, max_read)
    n = len(read_list)
    cut = np.array([int(np.ceil(0.3 * len(r))) for r in read_list], dtype=np.int32)
    tab = max_dna if scheme == 1 else max_read + max_read // 2 + 8
    return s3o.dp_align(dna, [len(d) for d in dna_list], max_dna, tab, rd, [len(r) for r in read_list], max_read, cut, scheme=scheme, scores=SC, **kw)


def test_known_answers_from_survey():
    rng = np.random.default_rng(3)
    win = rng.integers(0, 4, size=140, dtype=np.uint8)
    exact = win[20:120].copy()
    sub = exact.copy(); sub[50] = (sub[50] + 1) % 4
    dele = np.concatenate([win[20:60], win[63:120]])            # 97 bases, 3 reference bases missing from the read
    ins = np.concatenate([win[20:60], np.array([(win[60] + 1) % 4, (win[61] + 2) % 4], dtype=np.uint8), win[60:118]])
    sc, hit, cnt, pat, _, _ = run([win] * 4, [exact, sub, dele, ins], 140, 100)
    assert list(sc) == [100, 97, 92, 94]
    assert list(hit) == [20, 20, 20, 20]                         # start column (0-based offset of the first aligned reference base)
    assert s3o.dp_pattern_ops(pat[0]) == [("M", 100)]
    assert [o for o, c in s3o.dp_pattern_ops(pat[1])].count("m") == 1
    assert ("D", 3) in s3o.dp_pattern_ops(pat[2])
    assert ("I", 2) in s3o.dp_pattern_ops(pat[3])
    assert list(cnt) == [1, 1, 1, 1]


def score_from_ops(ops):
    s = 0
    for op, c in ops:
        if op == "M":
            s += c * SC[0]
        elif op == "m":
            s += c * SC[1]
        elif op in "DI":
            s += SC[2] + (c - 1) * SC[3]
    return s


def test_score_equals_score_of_emitted_ops():
    P = make_problems(96, 152, 400, seed=5, clip=True, anchors=False)
    for scheme in (1, 2):
        tab = 400 if scheme == 1 else 152 + 76 + 8
        sc, hit, cnt, pat, crt, start = s3o.dp_align(P["dna"], P["dna_len"], 400, tab, P["reads"], P["read_len"], 152, P["cutoffs"], scheme=scheme,
                                                     scores=SC, clip_lt=P["clip_lt"], clip_rt=P["clip_rt"])
        checked = 0
        for t in range(len(sc)):
            if sc[t] < P["cutoffs"][t]:
                continue
            ops = s3o.dp_pattern_ops(pat[t])
            assert sum(c for o, c in ops if o in "MmIS") == P["read_len"][t]
            # a leading insertion next to the left clip is charged as a free clip extension by the recurrence; skip those
            if any(o == "I" for o, c in ops[:2]) and ops[0][0] == "S":
                continue
            assert score_from_ops(ops) == sc[t], (t, ops, sc[t])
            checked += 1
        assert checked > 20
