"""ctypes binding of the CPU oracle (oracle/s3_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libs3oracle.so")


def build(force=False):
    src = os.path.join(ROOT, "oracle", "s3_oracle.c")
    src2 = os.path.join(ROOT, "oracle", "s3_oracle_dp.c")
    src3 = os.path.join(ROOT, "oracle", "s3_oracle_pe.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(x) for x in (src, src2, src3)):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, src, src2, src3, "-lm"])
    return LIB


class Step(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("type", "start", "end", "ceThreshold", "ceStart", "ceEnd", "minErr", "maxErr", "dir", "len")]


class Case(C.Structure):
    _fields_ = [("id", C.c_int), ("nSteps", C.c_int), ("steps", Step * 5), ("leftMost", C.c_int16 * 1024)]


class Model(C.Structure):
    _fields_ = [("readLength", C.c_int), ("maxError", C.c_int), ("nCases", C.c_int), ("cases", Case * 10)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rankQueries", "ltLookups", "saReads", "ceWords", "hits")]


HIT_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("kind", "u1"), ("strand", "u1"), ("nmis", "u1"), ("cas", "u1")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.s3o_index_load.restype = C.c_void_p
        _lib.s3o_index_load.argtypes = [C.c_char_p]
        _lib.s3o_index_free.argtypes = [C.c_void_p]
        _lib.s3o_model_build.argtypes = [C.c_int, C.c_int, C.POINTER(Model)]
        _lib.s3o_search_batch.restype = C.c_long
        _lib.s3o_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        _lib.s3o_decode.restype = C.c_int
        _lib.s3o_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Stats)]
    return _lib


def model(L, e):
    m = Model()
    lib().s3o_model_build(L, e, C.byref(m))
    return m


def model_lma_hash(case):
    h = 1469598103934665603
    for s in range(case.nSteps):
        st = case.steps[s]
        for q in range(st.len):
            pos = st.start + st.dir * q
            h = ((h ^ (case.leftMost[pos] & 0xFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def model_lines(L, e):
    """same text form as `ref_harness models`"""
    m = model(L, e)
    out = []
    for c in range(m.nCases):
        cs = m.cases[c]
        s = "L %d E %d C %d id %d :" % (L, e, c, cs.id)
        for k in range(cs.nSteps):
            st = cs.steps[k]
            err = 0 if st.type in (2, 4) else 1
            s += " [%d %d %d ce %d %d %d err %d %d %d]" % (st.type, st.start, st.end, st.ceThreshold, st.ceStart, st.ceEnd,
                                                           err, st.minErr, st.maxErr)
        s += " lma %016x" % model_lma_hash(cs)
        out.append(s)
    return out


class Index:
    def __init__(self, prefix=None, host_index=None):
        if host_index is not None:
            hi = host_index
            u32p = lambda a: a.ctypes.data_as(C.c_void_p)
            cum = (C.c_uint32 * 5)(*hi.fwd["cum"])
            rcum = (C.c_uint32 * 5)(*hi.rev["cum"])
            f = lib().s3o_index_create
            f.restype = C.c_void_p
            f.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p,
                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
            self.h = f(hi.n, u32p(hi.fwd["bwt"]), len(hi.fwd["bwt"]), u32p(hi.fwd["occ"]), hi.fwd["inverseSa0"], cum,
                       u32p(hi.rev["bwt"]), len(hi.rev["bwt"]), u32p(hi.rev["occ"]), hi.rev["inverseSa0"], rcum,
                       u32p(hi.sa), u32p(hi.pac), len(hi.pac), u32p(hi.lkt), u32p(hi.rev_lkt), hi.lkt_size)
        else:
            self.h = lib().s3o_index_load(prefix.encode())
        if not self.h:
            raise RuntimeError("cannot load index")

    def close(self):
        if self.h:
            lib().s3o_index_free(self.h)
            self.h = None

    def search(self, queries, read_lengths, wpq, mm, max_out=1000, hit_cap=None):
        n = len(read_lengths)
        queries = np.ascontiguousarray(queries, dtype=np.uint32)
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        if hit_cap is None:
            hit_cap = max(1024, 64 * n)
        while True:
            hits = np.zeros(hit_cap, dtype=HIT_DTYPE)
            offs = np.zeros(n + 1, dtype=np.uint64)
            tot = np.zeros(n, dtype=np.uint32)
            werr = np.zeros((n, 8), dtype=np.uint32)
            st = Stats()
            rc = lib().s3o_search_batch(self.h, queries.ctypes.data, read_lengths.ctypes.data, n, wpq, mm, max_out,
                                        hits.ctypes.data, hit_cap, offs.ctypes.data, tot.ctypes.data, werr.ctypes.data, C.byref(st))
            if rc >= 0:
                return hits[:rc], offs, tot, werr, st
            hit_cap *= 4

    def align_single_allbest(self, queries, read_lengths, wpq, mm, max_out=1000, occ_cap=None, first=0, count=None, report_type=2, seed_lengths=None):
        n_all = len(read_lengths)
        queries = np.ascontiguousarray(queries, dtype=np.uint32)
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        n = n_all if count is None else count
        assert first % 32 == 0
        f = lib().s3o_align_single_long
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Stats)]
        if occ_cap is None:
            occ_cap = max(4096, 32 * n)
        qptr = queries.ctypes.data + first * wpq * 4
        lptr = read_lengths.ctypes.data + first * 4
        sptr = None
        if seed_lengths is not None:
            seed_lengths = np.ascontiguousarray(seed_lengths, dtype=np.uint32)
            sptr = seed_lengths.ctypes.data + first * 4
        while True:
            phase = np.zeros(n, dtype=np.uint8)
            werr = np.zeros((n, 8), dtype=np.uint32)
            offs = np.zeros(n + 1, dtype=np.uint64)
            pos = np.zeros(occ_cap, dtype=np.uint32)
            strand = np.zeros(occ_cap, dtype=np.uint8)
            nmis = np.zeros(occ_cap, dtype=np.uint8)
            st = Stats()
            rc = f(self.h, qptr, lptr, sptr, n, wpq, mm, max_out, report_type, phase.ctypes.data, werr.ctypes.data, offs.ctypes.data,
                   pos.ctypes.data, strand.ctypes.data, nmis.ctypes.data, occ_cap, C.byref(st))
            if rc >= 0:
                return phase, werr, offs, pos[:rc], strand[:rc], nmis[:rc], st
            occ_cap *= 4

    def align_pair(self, queries, read_lengths, wpq, mm, max_hits, report_type, insert_low, insert_high, pe_max_output=1000):
        """paired-end ladder (oracle/s3_oracle_pe.c); reads 2p / 2p+1 are a pair.  Returns the arrays of s3dp.Batch.fetch_pairs."""
        import s3dp
        queries = np.ascontiguousarray(queries, dtype=np.uint32)
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        n = len(read_lengths)
        f = lib().s3o_align_pair
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Stats)]
        xcap = ocap = max(4096, 64 * n)
        while True:
            pairs = np.zeros(n // 2, dtype=s3dp.PAIR_DTYPE)
            xoff = np.zeros(n // 2 + 1, dtype=np.uint64)
            xa = np.zeros(xcap, dtype=s3dp.PAIR_XA_DTYPE)
            ooff = np.zeros(n + 1, dtype=np.uint64)
            pos = np.zeros(ocap, dtype=np.uint32)
            strand = np.zeros(ocap, dtype=np.uint8)
            mis = np.zeros(ocap, dtype=np.uint8)
            st = Stats()
            rc = f(self.h, queries.ctypes.data, read_lengths.ctypes.data, n, wpq, mm, max_hits, report_type, insert_low, insert_high, pe_max_output,
                   pairs.ctypes.data, xoff.ctypes.data, xa.ctypes.data, xcap, ooff.ctypes.data, pos.ctypes.data, strand.ctypes.data, mis.ctypes.data,
                   ocap, C.byref(st))
            if rc == 0:
                nx, no = int(xoff[-1]), int(ooff[-1])
                return pairs, xoff, xa[:nx], ooff, pos[:no], strand[:no], mis[:no]
            xcap *= 4
            ocap *= 4

    def decode(self, hits):
        cap = int(sum(1 if h["kind"] == 1 else int(h["b"]) - int(h["a"]) + 1 for h in hits)) + 1
        pos = np.zeros(cap, dtype=np.uint32)
        strand = np.zeros(cap, dtype=np.uint8)
        nmis = np.zeros(cap, dtype=np.uint8)
        hh = np.ascontiguousarray(hits)
        k = lib().s3o_decode(self.h, hh.ctypes.data, len(hh), pos.ctypes.data, strand.ctypes.data, nmis.ctypes.data, cap, None)
        return pos[:k], strand[:k], nmis[:k]


CODE = np.full(256, 2, dtype=np.uint8)  # non-ACGT -> G (IndexHandler.cpp:44-48)
for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    CODE[ch] = v


def word_per_query(max_read_length):
    w = 1
    while w < max_read_length:
        w *= 2
    return w // 16


def pack_reads(reads, max_read_length):
    """reads: list of bytes.  Reference layout: 32-interleaved words, 16 chars/word LSB first (QueryParser.cpp, DV-Kernel.cu:4268)."""
    n = len(reads)
    wpq = word_per_query(max_read_length)
    round_up = (n + 31) // 32 * 32
    q = np.zeros(round_up * wpq, dtype=np.uint32)
    lens = np.zeros(round_up, dtype=np.uint32)
    for r, s in enumerate(reads):
        codes = CODE[np.frombuffer(s, dtype=np.uint8)].astype(np.uint32)
        L = len(codes)
        lens[r] = L
        pad = np.zeros(wpq * 16, dtype=np.uint32)
        pad[:L] = codes
        words = (pad.reshape(wpq, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
        base = r // 32 * 32 * wpq + r % 32
        q[base + 32 * np.arange(wpq)] = words
    return q, lens[:n], wpq


def read_fastq(path):
    out = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith(b"@"):
            out.append(lines[i + 1].strip())
            i += 4
        elif lines[i].startswith(b">"):
            out.append(lines[i + 1].strip())
            i += 2
        else:
            i += 1
    return out


# ---- DP oracle (oracle/s3_oracle_dp.c; pinned on the GPU box against the reference's own device code, see its header) ----
class DpScoresO(C.Structure):
    _fields_ = [("match", C.c_int), ("mismatch", C.c_int), ("gapOpen", C.c_int), ("gapExtend", C.c_int)]


def dp_align(packed_dna, dna_lengths, max_dna, max_table, packed_reads, read_lengths, max_read, cutoffs, scheme=1, scores=(1, -2, -3, -1),
             clip_lt=None, clip_rt=None, anchor_lt=None, anchor_rt=None):
    n = len(read_lengths)
    u32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint32)
    packed_dna, dna_lengths, packed_reads, read_lengths = u32(packed_dna), u32(dna_lengths), u32(packed_reads), u32(read_lengths)
    clip_lt, anchor_lt, anchor_rt = u32(clip_lt), u32(anchor_lt), u32(anchor_rt)
    clip_rt = None if clip_rt is None else np.array(clip_rt, dtype=np.uint32)
    cutoffs = np.ascontiguousarray(cutoffs, dtype=np.int32)
    pat_len = max_read + max_table
    sc = np.zeros(n, dtype=np.int32)
    hit = np.zeros(n, dtype=np.uint32)
    start = np.zeros(n, dtype=np.uint32)
    cnt = np.zeros(n, dtype=np.uint32)
    pat = np.zeros((n, pat_len), dtype=np.uint8)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    f = lib().s3o_dp_align
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.POINTER(DpScoresO), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    s = DpScoresO(*scores)
    f(p(packed_dna), p(dna_lengths), max_dna, max_table, p(packed_reads), p(read_lengths), max_read, n, p(clip_lt), p(clip_rt), p(anchor_lt), p(anchor_rt),
      C.byref(s), p(cutoffs), scheme, p(sc), p(hit), p(start), p(cnt), p(pat), pat_len)
    return sc, hit, cnt, pat, clip_rt, start


def dp_pattern_ops(pat_row):
    """decode the reversed op string (M m D I S + V<n> run markers) into a forward list of (op, count)"""
    ops = []
    k = 0
    raw = []
    while k < len(pat_row) and pat_row[k] != 0:
        ch = chr(pat_row[k])
        if k + 2 < len(pat_row) + 1 and k + 1 < len(pat_row) and chr(pat_row[k + 1]) == 'V' and ch in "SI":
            raw.append((ch, int(pat_row[k + 2])))
            k += 3
        else:
            raw.append((ch, 1))
            k += 1
    for op, cnt in reversed(raw):
        if ops and ops[-1][0] == op:
            ops[-1] = (op, ops[-1][1] + cnt)
        else:
            ops.append((op, cnt))
    return ops
