"""s3dp -- thin ctypes binding of libs3dp.so (include/s3dp.h) used by tests and bench.py.

The product is the native library; this module only moves numpy arrays across the C-ABI and
reads the 2BWT / GPU-2BWT index files.  There is no Python or CPU implementation of the search
here: if the library or a HIP device is missing the calls raise.
"""
import ctypes as C
import os
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))          # soap3-dp_amd/
LIB_PATH = os.path.join(PKG_ROOT, "libs3dp.so")

HIT_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("kind", "u1"), ("strand", "u1"), ("nmis", "u1"), ("cas", "u1")])


class S3Error(RuntimeError):
    pass


class BwtDesc(C.Structure):
    _fields_ = [("bwt", C.c_void_p), ("bwtWords", C.c_size_t), ("occ", C.c_void_p), ("occRecs", C.c_size_t),
                ("inverseSa0", C.c_uint32), ("cumFreq", C.c_uint32 * 5)]


class IndexDesc(C.Structure):
    _fields_ = [("fwd", BwtDesc), ("rev", BwtDesc), ("textLength", C.c_uint32), ("sa", C.c_void_p),
                ("packedDNA", C.c_void_p), ("packedWords", C.c_size_t), ("lkt", C.c_void_p), ("revLkt", C.c_void_p),
                ("lktSize", C.c_int), ("saInterval", C.c_uint32)]


class StreamResult(C.Structure):
    _fields_ = [("ticket", C.c_uint64), ("n", C.c_uint32), ("phase", C.POINTER(C.c_uint8)), ("withError", C.POINTER(C.c_uint32)),
                ("occOffsets", C.POINTER(C.c_uint32)), ("occPos", C.POINTER(C.c_uint32)), ("occMeta", C.POINTER(C.c_uint32)),
                ("deviceMs", C.c_float), ("batch", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rankQueries", "ltLookups", "saReads", "ceWindows", "records")]


class KernelStats(C.Structure):
    _fields_ = [("ms", C.c_float), ("launches", C.c_uint32), ("rankQueries", C.c_uint64), ("ltLookups", C.c_uint64), ("saReads", C.c_uint64), ("ceWindows", C.c_uint64),
                ("listedRankQueries", C.c_uint64), ("accepted", C.c_uint64), ("ceRejected", C.c_uint64 * 18)]


EXPORTS = ["s3_index_upload", "s3_index_free", "s3_index_device_bytes", "s3_index_lookup_chars", "s3_index_flank_tables", "s3_index_fetch_flank", "s3_batch_create", "s3_batch_free", "s3_batch_upload",
           "s3_batch_search", "s3_batch_fetch_hits", "s3_batch_align_single_allbest", "s3_batch_align_single", "s3_batch_set_seed_lengths", "s3_batch_align_pair", "s3_batch_fetch_pairs", "s3_batch_fetch_pair_origins", "s3_batch_route_pairs", "s3_batch_no_pair_result", "s3_batch_fetch_routing", "s3_batch_fetch_kept", "s3_dp_new_default_routed", "s3_dp_stage_parameters", "s3_seed_positions", "s3_dp_half_end", "s3_seed_candidates", "s3_dp_single_end", "s3_dp_deep", "s3_dp_new_default", "s3_batch_fetch_alignments", "s3_batch_sync", "s3_batch_last_ce_timing",
           "s3_batch_last_timing", "s3_batch_set_grid", "s3_batch_kernel_stats", "s3_batch_counters", "s3_batch_enable_counters", "s3_bwt_search", "s3_free", "s3_model_describe",
           "s3_device_count", "s3_device_host_node", "s3_last_error", "s3_version", "s3_probe_gather", "s3_probe_copy", "s3_dp_totals",
           "s3_stream_create", "s3_stream_acquire", "s3_stream_submit", "s3_stream_next", "s3_stream_release", "s3_stream_free", "s3_host_staging_reserve", "s3_host_staging_release",
           "s3_group_create", "s3_group_size", "s3_group_device", "s3_group_index", "s3_group_member_of_batch", "s3_group_allreduce", "s3_group_free"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise S3Error("libs3dp.so is not built (run `make -C soap3-dp_amd` or __graft_entry__.build())")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # (see s3dp_api.hip: hardware queues for the streams of the batches in flight)
        L = C.CDLL(LIB_PATH)
        L.s3_last_error.restype = C.c_char_p
        L.s3_version.restype = C.c_char_p
        L.s3_index_device_bytes.restype = C.c_size_t
        L.s3_index_device_bytes.argtypes = [C.c_void_p]
        L.s3_index_upload.argtypes = [C.c_int, C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
        L.s3_index_free.argtypes = [C.c_void_p]
        L.s3_batch_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.s3_batch_free.argtypes = [C.c_void_p]
        L.s3_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.s3_batch_search.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.s3_batch_fetch_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.s3_batch_align_single_allbest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.s3_batch_fetch_alignments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.POINTER(C.c_uint64)]
        L.s3_batch_sync.argtypes = [C.c_void_p]
        L.s3_batch_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.s3_batch_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.s3_batch_enable_counters.argtypes = [C.c_void_p, C.c_int]
        L.s3_model_describe.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.s3_stream_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.s3_stream_acquire.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]
        L.s3_stream_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
        L.s3_stream_next.argtypes = [C.c_void_p, C.POINTER(StreamResult)]
        L.s3_stream_release.argtypes = [C.c_void_p]
        L.s3_stream_free.argtypes = [C.c_void_p]
        L.s3_stream_free.restype = None
        L.s3_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
        L.s3_group_size.argtypes = [C.c_void_p]
        L.s3_group_device.argtypes = [C.c_void_p, C.c_int]
        L.s3_group_index.argtypes = [C.c_void_p, C.c_int]
        L.s3_group_index.restype = C.c_void_p
        L.s3_group_member_of_batch.argtypes = [C.c_void_p, C.c_uint64]
        L.s3_group_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.s3_group_free.argtypes = [C.c_void_p]
        L.s3_group_free.restype = None
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise S3Error("libs3dp error %d: %s" % (rc, lib().s3_last_error().decode()))


def probe_gather(table_bytes, blocks=256 * 8, iters=256, ilp=4, loads_per_line=2, dependent=0, repeats=3, device=0):
    ms, lines = C.c_float(), C.c_uint64()
    f = lib().s3_probe_gather
    f.argtypes = [C.c_int, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    _chk(f(device, table_bytes, blocks, iters, ilp, loads_per_line, dependent, repeats, C.byref(ms), C.byref(lines)))
    return ms.value, lines.value


def host_staging_reserve(max_reads, word_per_query, depth, occ_per_read):
    """pinned host blocks for the staging of a stream's slots, made ahead of time (include/s3dp.h: s3_host_staging_reserve)"""
    f = lib().s3_host_staging_reserve
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    _chk(f(max_reads, word_per_query, depth, occ_per_read))


def host_staging_release():
    f = lib().s3_host_staging_release
    f.restype = None
    f()


def probe_copy(nbytes, repeats=4, device=0):
    """average milliseconds of one device-to-device copy of nbytes with the library's own 16-byte copy kernel"""
    ms = C.c_float()
    f = lib().s3_probe_copy
    f.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    _chk(f(device, nbytes, repeats, C.byref(ms)))
    return ms.value


def device_count():
    return lib().s3_device_count()


def model_text(L, e):
    buf = C.create_string_buffer(8192)
    n = lib().s3_model_describe(L, e, buf, 8192)
    if n < 0:
        _chk(n)
    return buf.value.decode()


# ------------------------------------------------------------------------------------------------
# index files
# ------------------------------------------------------------------------------------------------
def build_lookup_table(codes, table_size=13):
    """cumulative k-mer counts with the text padded by (k-1) 'A's, as 2bwt-flex/LTConstruct.c:45-101 writes them"""
    n = len(codes)
    padded = np.concatenate([codes.astype(np.uint32), np.zeros(table_size - 1, dtype=np.uint32)])
    v = np.zeros(n, dtype=np.uint32)
    for j in range(table_size):
        v = (v << np.uint32(2)) | padded[j:j + n]
    vals, counts = np.unique(v, return_counts=True)
    cnt = np.zeros(1 << (2 * table_size), dtype=np.uint32)
    cnt[vals] = counts.astype(np.uint32)
    return np.cumsum(cnt, dtype=np.uint32)


def unpack_pac(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = (len(raw) - 2) * 4 + int(raw[-1])
    data = raw[:-1]
    codes = np.empty(len(data) * 4, dtype=np.uint8)
    codes[0::4] = data >> 6
    codes[1::4] = (data >> 4) & 3
    codes[2::4] = (data >> 2) & 3
    codes[3::4] = data & 3
    return codes[:n]


def pack_text_words(codes):
    """16 chars per uint32, first char in the most significant bits, zero padded (TextConverter.c:666-720)"""
    n = len(codes)
    words = (n + 63) // 64 * 4 + 8
    padded = np.zeros(words * 16, dtype=np.uint32)
    padded[:n] = codes
    sh = (2 * (15 - np.arange(16, dtype=np.uint32))).astype(np.uint32)
    return (padded.reshape(words, 16) << sh).sum(axis=1, dtype=np.uint64).astype(np.uint32)


def pac_words_from_file(path, n=None):
    """the packed text words of pack_text_words() straight from a .pac file: its bytes hold 4 characters each, first one in the top bits, so four
    bytes read big-endian ARE one word of 16 characters (TextConverter.c:666-720); the unused bits of the last byte are cleared by the writer"""
    raw = np.fromfile(path, dtype=np.uint8)
    m = (len(raw) - 2) * 4 + int(raw[-1])
    if n is not None:
        assert m == n
    nwords = (m + 63) // 64 * 4 + 8
    buf = np.zeros(nwords * 4, dtype=np.uint8)
    k = (m + 3) // 4
    buf[:k] = raw[:k]
    if m % 4:
        buf[k - 1] &= (0xFF << (2 * (4 - m % 4))) & 0xFF
    return buf.view(">u4").astype(np.uint32)


def fmv_words(bwt_words, n, inverse_sa0, cum, use_gpu=None):
    """The words of a <prefix>.fmv / .rev.fmv file -- the occurrence arrays of the reference's CPU 2BWT (BWTGenerateOccValueFromBwt, 2bwt-lib/BWTConstruct.c:
    1037-1247; read back by BWTLoad, BWT.c:196-223): a sample every 256 BWT characters, 16 bits per symbol counted from the last MAJOR sample (every 65,536
    characters, 32 bits, absolute), two samples per word -- sample 2j in the high half, 2j + 1 in the low half --, the minor words first, then the majors.
    The product never reads these files (its kernels use the .fmv.gpu layout); they exist so that the REFERENCE's own host code (oracle/_ref/ref_harness) can
    load an index this builder wrote -- bench.py's DP parity leg at the benchmark's size."""
    n = int(n)
    N = (n + 255) // 256 + 1                                       # samples (one at either end)
    need = (N - 1) * 16
    w = np.zeros(need, dtype=np.uint32)
    m = min(len(bwt_words), need)
    w[:m] = np.asarray(bwt_words[:m], dtype=np.uint32)
    lut = np.zeros((256, 4), dtype=np.uint32)                      # symbols of a byte's four characters (their order inside the byte is no matter)
    for v in range(256):
        for k in range(4):
            lut[v, (v >> (2 * k)) & 3] += 1
    b = w.view(np.uint8)
    cnt = np.zeros((N - 1, 4), dtype=np.uint64)
    dev = None
    if use_gpu or (use_gpu is None and N > (1 << 20)):             # (a genome-scale text: the counting on the GPU when there is one -- 30 s of numpy otherwise)
        try:
            import torch
            dev = torch.device("cuda") if torch.cuda.is_available() else None
        except ImportError:
            dev = None
    step = 1 << 18                                                 # intervals per piece (64 bytes each)
    if dev is not None:
        import torch
        tl = torch.from_numpy(lut.astype(np.int32)).to(dev)
    for i0 in range(0, N - 1, step):
        i1 = min(N - 1, i0 + step)
        if dev is not None:
            tb = torch.from_numpy(b[i0 * 64:i1 * 64]).to(dev).long()
            cnt[i0:i1] = tl[tb].reshape(i1 - i0, 64, 4).sum(dim=1).cpu().numpy().astype(np.uint64)
        else:
            cnt[i0:i1] = lut[b[i0 * 64:i1 * 64]].reshape(i1 - i0, 64, 4).sum(axis=1)
    absol = np.zeros((N, 4), dtype=np.uint64)
    np.cumsum(cnt, axis=0, out=absol[1:])
    n_major = (N + 255) // 256
    major = absol[0:N:256][:n_major]
    rel = (absol - major[np.arange(N) // 256]).astype(np.uint32)   # < 65,536: a sample on a major boundary counts from itself
    W = (N + 1) // 2
    hi = rel[0::2][:W]
    lo = rel[1::2]
    if len(lo) < W:                                                # an odd number of samples: the last word's low half repeats its high half
        lo = np.concatenate([lo, rel[N - 1:N]])
    occ = ((hi << np.uint32(16)) | lo).astype(np.uint32)
    head = np.array([int(inverse_sa0)] + [int(x) for x in cum[1:5]], dtype=np.uint32)
    return np.concatenate([head, occ.ravel(), major.astype(np.uint32).ravel()])


class HostIndex:
    """host arrays of one 2BWT + GPU-2BWT index in the file layouts"""

    def __init__(self):
        self.keep = []

    @classmethod
    def from_files(cls, prefix, lkt=None, rev_lkt=None, shared=False):
        """shared=True: the big arrays are private copy-on-write mappings of the files (np.memmap mode 'c') instead of copies in this process's
        heap -- N ranks of one node that load the same files then share ONE set of pages in the page cache (a GRCh38-scale index is 17 GB per
        private copy), as the reference's share-index mode does (IndexHandler.cpp:176-192); the packed text comes from the .pac bytes directly and
        the unpacked characters (`codes`, which only builders and tests use) are not materialised."""
        self = cls()

        def words(path):
            return np.memmap(path, dtype=np.uint32, mode="c") if shared else np.fromfile(path, dtype=np.uint32)

        def half(mid):
            raw = words(prefix + mid + ".bwt")
            occ = words(prefix + mid + ".fmv.gpu")
            assert (raw[:5] == occ[:5]).all(), "header mismatch between .bwt and .fmv.gpu"
            return dict(inverseSa0=int(raw[0]), cum=[0] + [int(x) for x in raw[1:5]], bwt=raw[5:] if shared else np.ascontiguousarray(raw[5:]),
                        occ=occ[5:] if shared else np.ascontiguousarray(occ[5:]))
        self.fwd = half("")
        self.rev = half(".rev")
        self.n = self.fwd["cum"][4]
        sa = words(prefix + ".sa")
        self.sa_interval = int(sa[5])                            # SaValueFreq: 1 = full array, k = every k-th entry (BWT.c:254-255)
        self.sa = sa[6:6 + (self.n + self.sa_interval) // self.sa_interval]
        if not shared:
            self.sa = np.ascontiguousarray(self.sa).copy()
        self.sa[0] = 0xFFFFFFFF                                   # BWT.c:286 (shared: the one page this touches becomes private)
        if shared:
            self.codes = None
            self.pac = pac_words_from_file(prefix + ".pac", self.n)
        else:
            self.codes = unpack_pac(prefix + ".pac")
            assert len(self.codes) == self.n
            self.pac = pack_text_words(self.codes)
        if lkt is None:
            if os.path.exists(prefix + ".lkt"):
                lkt = words(prefix + ".lkt")[1:]
                rev_lkt = words(prefix + ".rev.lkt")[1:]
                if len(lkt) != 1 << 26 or len(rev_lkt) != 1 << 26:
                    raise ValueError("%s.lkt / .rev.lkt are not 13-mer tables of 2^26 entries" % prefix)
            else:
                codes = self.codes if self.codes is not None else unpack_pac(prefix + ".pac")
                lkt = build_lookup_table(codes)
                rev_lkt = build_lookup_table(codes[::-1])
        self.lkt = lkt if shared else np.ascontiguousarray(lkt)
        self.rev_lkt = rev_lkt if shared else np.ascontiguousarray(rev_lkt)
        self.lkt_size = 13
        return self

    def write_fmv_files(self, prefix):
        """<prefix>.fmv and .rev.fmv beside files written by write_files: the CPU 2BWT's occurrence arrays (fmv_words), which only the reference's own
        host code reads"""
        for mid, h in (("", self.fwd), (".rev", self.rev)):
            fmv_words(h["bwt"], self.n, h["inverseSa0"], h["cum"]).astype("<u4").tofile(prefix + mid + ".fmv")

    def write_files(self, prefix, name="chr1", with_lkt=False):
        """Write the reference's index files (<prefix>.bwt .fmv.gpu .rev.bwt .rev.fmv.gpu .sa .pac .ann .amb .tra, optionally .lkt / .rev.lkt)
        for a genome that is ONE sequence without N-runs (what bench / tools build on the GPU).  Layouts: 2bwt-lib/BWT.c:169-194,286
        (.bwt, .sa), BGS-Build.cpp:139-192 (.fmv.gpu), TextConverter.c:666-720 (.pac), HSP.c:256-531 (.ann, .amb, .tra)."""
        n = int(self.n)

        def words(path, header, payload):
            with open(path, "wb") as f:
                np.asarray(header, dtype="<u4").tofile(f)
                np.ascontiguousarray(payload, dtype="<u4").tofile(f)

        for mid, h in (("", self.fwd), (".rev", self.rev)):
            hdr = [h["inverseSa0"]] + [int(x) for x in h["cum"][1:5]]
            words(prefix + mid + ".bwt", hdr, h["bwt"])
            words(prefix + mid + ".fmv.gpu", hdr, h["occ"])
        sa_file = np.array(self.sa, dtype=np.uint32, copy=True)
        sa_file[0] = n                                            # on disk entry 0 is the suffix '$' = n; readers force it to -1 (BWT.c:286)
        words(prefix + ".sa", [self.fwd["inverseSa0"]] + [int(x) for x in self.fwd["cum"][1:5]] + [getattr(self, "sa_interval", 1)], sa_file)
        # packed text: 4 characters per byte, first one in the top bits; then one (partial or empty) byte and the count n % 4
        pw = np.ascontiguousarray(self.pac, dtype=">u4")
        raw = pw.view(np.uint8)[: n // 4 + 1].copy()
        if n % 4:
            raw[-1] &= (0xFF << (2 * (4 - n % 4))) & 0xFF
        else:
            raw[-1] = 0
        with open(prefix + ".pac", "wb") as f:
            raw.tofile(f)
            f.write(bytes([n % 4]))
        with open(prefix + ".ann", "w") as f:
            f.write("%u 1 %u\n0 %s\n0 %u 0\n" % (n, 11, name, n))
        with open(prefix + ".amb", "w") as f:
            f.write("%u 1 0\n" % n)
        grid = (n >> 18) + 1
        with open(prefix + ".tra", "w") as f:
            f.write("%u 1 0 %u\n" % (n, grid))
            f.write("0\n" * grid)
            f.write("0 1 4294967295\n")
            f.write("0 %u\n" % n)
        if with_lkt:
            words(prefix + ".lkt", [13], self.lkt)
            words(prefix + ".rev.lkt", [13], self.rev_lkt)

    def desc(self):
        d = IndexDesc()
        for name, h in (("fwd", self.fwd), ("rev", self.rev)):
            b = getattr(d, name)
            b.bwt = h["bwt"].ctypes.data
            b.bwtWords = len(h["bwt"])
            b.occ = h["occ"].ctypes.data
            b.occRecs = len(h["occ"]) // 4
            b.inverseSa0 = h["inverseSa0"]
            for i in range(5):
                b.cumFreq[i] = h["cum"][i]
        d.textLength = self.n
        d.sa = self.sa.ctypes.data
        d.saInterval = getattr(self, "sa_interval", 1)
        d.packedDNA = self.pac.ctypes.data
        d.packedWords = len(self.pac)
        d.lkt = self.lkt.ctypes.data
        d.revLkt = self.rev_lkt.ctypes.data
        d.lktSize = self.lkt_size
        return d


class Index:
    def __init__(self, host_index, device=0):
        self.h = C.c_void_p()
        d = host_index.desc()
        _chk(lib().s3_index_upload(device, C.byref(d), C.byref(self.h)))

    def device_bytes(self):
        return lib().s3_index_device_bytes(self.h)

    def flank_tables(self):
        """0, 1 (forward index only) or 2: the flank tables of check-and-extend the upload found room for"""
        f = lib().s3_index_flank_tables
        f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def fetch_flank(self, table, first, count):
        """entries [first, first + count) of flank table 0 (forward index order) / 1 (reverse index order): uint32 [count, 4] = position, 3 x 16 characters"""
        out = np.zeros((count, 4), dtype=np.uint32)
        f = lib().s3_index_fetch_flank
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        _chk(f(self.h, int(table), int(first), int(count), out.ctypes.data))
        return out

    def close(self):
        if self.h:
            lib().s3_index_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    def __init__(self, index, max_reads, wpq):
        self.index = index
        self.h = C.c_void_p()
        self.n = 0
        self.wpq = wpq
        _chk(lib().s3_batch_create(index.h, max_reads, wpq, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().s3_batch_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, queries, read_lengths):
        queries = np.ascontiguousarray(queries, dtype=np.uint32)
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        self.n = len(read_lengths)
        need = (self.n + 31) // 32 * 32 * self.wpq              # s3_batch_upload reads whole groups of 32 reads (include/s3dp.h)
        if queries.size < need:
            queries = np.concatenate([queries.ravel(), np.zeros(need - queries.size, dtype=np.uint32)])
        _chk(lib().s3_batch_upload(self.h, queries.ctypes.data, read_lengths.ctypes.data, self.n))

    def set_grid(self, blocks_per_cu):
        """width of this batch's persistent search grids in workgroups per compute unit (0: all that are resident at once) -- s3_batch_set_grid"""
        f = lib().s3_batch_set_grid
        f.argtypes = [C.c_void_p, C.c_uint]
        _chk(f(self.h, int(blocks_per_cu)))

    def enable_counters(self, on=True):
        _chk(lib().s3_batch_enable_counters(self.h, 1 if on else 0))

    def search(self, mm, max_out=1000):
        _chk(lib().s3_batch_search(self.h, mm, max_out))

    def sync(self):
        _chk(lib().s3_batch_sync(self.h))

    def fetch_hits(self):
        n = self.n
        offs = np.zeros(n + 1, dtype=np.uint64)
        tot = np.zeros(n, dtype=np.uint32)
        werr = np.zeros((n, 8), dtype=np.uint32)
        nh = C.c_uint64()
        _chk(lib().s3_batch_fetch_hits(self.h, offs.ctypes.data, None, 0, tot.ctypes.data, werr.ctypes.data, C.byref(nh)))
        hits = np.zeros(max(1, nh.value), dtype=HIT_DTYPE)
        _chk(lib().s3_batch_fetch_hits(self.h, None, hits.ctypes.data, len(hits), None, None, None))
        return hits[:nh.value], offs, tot, werr

    def align_single_allbest(self, mm, max_out=1000):
        _chk(lib().s3_batch_align_single_allbest(self.h, mm, max_out))

    def set_seed_lengths(self, seed_lengths):
        """long-read mode (reference: seedLengths of all_valid_alignment); None switches it off"""
        f = lib().s3_batch_set_seed_lengths
        f.argtypes = [C.c_void_p, C.c_void_p]
        if seed_lengths is None:
            _chk(f(self.h, None))
            return
        a = np.ascontiguousarray(seed_lengths, dtype=np.uint32)
        _chk(f(self.h, a.ctypes.data))

    def align_single(self, mm, max_out=1000, report_type=2):
        f = lib().s3_batch_align_single
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        _chk(f(self.h, mm, max_out, report_type))

    def fetch_alignments(self):
        n = self.n
        phase = np.zeros(n, dtype=np.uint8)
        werr = np.zeros((n, 8), dtype=np.uint32)
        offs = np.zeros(n + 1, dtype=np.uint64)
        no = C.c_uint64()
        _chk(lib().s3_batch_fetch_alignments(self.h, phase.ctypes.data, werr.ctypes.data, offs.ctypes.data, None, None, None, 0, C.byref(no)))
        k = max(1, no.value)
        pos = np.zeros(k, dtype=np.uint32)
        strand = np.zeros(k, dtype=np.uint8)
        mis = np.zeros(k, dtype=np.uint8)
        _chk(lib().s3_batch_fetch_alignments(self.h, None, None, None, pos.ctypes.data, strand.ctypes.data, mis.ctypes.data, k, None))
        return phase, werr, offs, pos[:no.value], strand[:no.value], mis[:no.value]

    def seed_candidates(self, seed_batch, read_lengths, stage, max_hit_num, read_ids=None):
        """seeding for the DP stages: returns (candidates SEED_CAND_DTYPE sorted by strand/read/position, per-seed status)"""
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        f = lib().s3_seed_candidates
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        cand, n, st, ns = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        ids = None if read_ids is None else np.ascontiguousarray(read_ids, dtype=np.uint32)
        _chk(f(self.h, seed_batch.h, read_lengths.ctypes.data, None if ids is None else ids.ctypes.data, 0 if ids is None else len(ids), stage, max_hit_num,
               C.byref(cand), C.byref(n), C.byref(st), C.byref(ns)))
        out = np.frombuffer(C.string_at(cand, n.value * SEED_CAND_DTYPE.itemsize), dtype=SEED_CAND_DTYPE).copy() if n.value else np.zeros(0, SEED_CAND_DTYPE)
        status = np.frombuffer(C.string_at(st, ns.value), dtype=np.uint8).copy() if ns.value else np.zeros(0, np.uint8)
        free = lib().s3_free
        free.argtypes = [C.c_void_p]
        free(cand)
        free(st)
        return out, status

    def align_pair(self, mm, max_hits_per_end, report_type, insert_low, insert_high, pe_max_output=1000):
        """reads 2p / 2p+1 are the ends of pair p; paired-end ladder with the DP stages off"""
        f = lib().s3_batch_align_pair
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        _chk(f(self.h, mm, max_hits_per_end, report_type, insert_low, insert_high, pe_max_output))

    def fetch_pairs(self):
        f = lib().s3_batch_fetch_pairs
        f.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
        n, npairs = self.n, self.n // 2
        pairs = np.zeros(npairs, dtype=PAIR_DTYPE)
        xoff = np.zeros(npairs + 1, dtype=np.uint64)
        ooff = np.zeros(n + 1, dtype=np.uint64)
        nx, no = C.c_uint64(), C.c_uint64()
        _chk(f(self.h, pairs.ctypes.data, xoff.ctypes.data, None, 0, C.byref(nx), ooff.ctypes.data, None, None, None, 0, C.byref(no)))
        xa = np.zeros(max(1, nx.value), dtype=PAIR_XA_DTYPE)
        pos = np.zeros(max(1, no.value), dtype=np.uint32)
        strand = np.zeros(max(1, no.value), dtype=np.uint8)
        mis = np.zeros(max(1, no.value), dtype=np.uint8)
        _chk(f(self.h, None, None, xa.ctypes.data, len(xa), None, None, pos.ctypes.data, strand.ctypes.data, mis.ctypes.data, len(pos), None))
        return pairs, xoff, xa[:nx.value], ooff, pos[:no.value], strand[:no.value], mis[:no.value]

    def timing(self):
        t, s, k = C.c_float(), C.c_float(), C.c_uint32()
        _chk(lib().s3_batch_last_timing(self.h, C.byref(t), C.byref(s), C.byref(k)))
        return t.value, s.value, k.value

    def ce_timing(self):
        """(ms inside the check-and-extend kernels of the last call, launches)"""
        s, k = C.c_float(), C.c_uint32()
        f = lib().s3_batch_last_ce_timing
        f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        _chk(f(self.h, C.byref(s), C.byref(k)))
        return s.value, k.value

    def kernel_stats(self):
        """per kernel of the last call's search passes (first pass, trunk pass, unit pass, check-and-extend): device ms, launches and, with counters
        enabled, its own rank queries / lookups / suffix-array reads / compare windows"""
        arr = (KernelStats * 4)()
        f = lib().s3_batch_kernel_stats
        f.argtypes = [C.c_void_p, C.POINTER(KernelStats)]
        _chk(f(self.h, arr))
        return {name: {k: (list(getattr(arr[i], k)) if k == "ceRejected" else getattr(arr[i], k)) for k, _ in KernelStats._fields_}
                for i, name in enumerate(("first", "trunk", "unit", "ce"))}

    def counters(self):
        c = Counters()
        _chk(lib().s3_batch_counters(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in Counters._fields_}


OCC_DTYPE = np.dtype([("readID", "<u4"), ("pos", "<u4"), ("strand", "u1"), ("mismatchCount", "u1"), ("pad", "u1", 2)])     # s3_occ
HALF_RESULT_DTYPE = np.dtype([("readID", "<u4"), ("algnmt_1", "<u4"), ("algnmt_2", "<u4"), ("score_1", "<i4"), ("score_2", "<i4"), ("strand_1", "u1"),
                              ("strand_2", "u1"), ("whichFromDP", "u1"), ("pad", "u1"), ("editdist", "<u4"), ("insertSize", "<u4"),
                              ("num_sameScore", "<u4"), ("cigarOffset", "<u4")])                                                 # s3_dp_half_result

DEEP_RESULT_DTYPE = np.dtype([("readID", "<u4"), ("algnmt_1", "<u4"), ("algnmt_2", "<u4"), ("score_1", "<i4"), ("score_2", "<i4"), ("editdist_1", "<u4"),
                              ("editdist_2", "<u4"), ("insertSize", "<u4"), ("num_sameScore_1", "<u4"), ("num_sameScore_2", "<u4"), ("cigarOffset_1", "<u4"),
                              ("cigarOffset_2", "<u4"), ("strand_1", "u1"), ("strand_2", "u1"), ("pad", "u1", 2)])                          # s3_dp_deep_result
SINGLE_RESULT_DTYPE = np.dtype([("readID", "<u4"), ("algnmt", "<u4"), ("score", "<i4"), ("editdist", "<u4"), ("num_sameScore", "<u4"),
                                ("cigarOffset", "<u4"), ("strand", "u1"), ("pad", "u1", 3)])                                             # s3_dp_single_result
SEED_CAND_DTYPE = np.dtype([("readID", "<u4"), ("pos", "<u4"), ("strand", "u1"), ("pad", "u1", 3)])                                    # s3_seed_candidate

# s3_pair_result / s3_pair_xa of include/s3dp.h
PAIR_DTYPE = np.dtype([("phase", "u1"), ("min1", "u1"), ("min2", "u1"), ("minTotal", "u1"), ("secMinTotal", "u1"), ("bestS1", "u1"), ("bestS2", "u1"),
                       ("bestM1", "u1"), ("bestM2", "u1"), ("pad", "u1", 3), ("x0_1", "<u4"), ("x1_1", "<u4"), ("x0_2", "<u4"), ("x1_2", "<u4"),
                       ("numPairs", "<u4"), ("numMin", "<u4"), ("bestA1", "<u4"), ("bestA2", "<u4")])
PAIR_XA_DTYPE = np.dtype([("a1", "<u4"), ("a2", "<u4"), ("s1", "u1"), ("s2", "u1"), ("m1", "u1"), ("m2", "u1")])

# ------------------------------------------------------------------------------------------------
# read packing (the layout loadSingleReadsGz produces; QueryParser.cpp:996, DV-Kernel.cu:4268-4270)
# ------------------------------------------------------------------------------------------------
CODE = np.full(256, 2, dtype=np.uint8)          # non-ACGT -> G (IndexHandler.cpp:44-48)
for _ch, _v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    CODE[_ch] = _v


def word_per_query(max_read_length):
    w = 1
    while w < max_read_length:
        w *= 2
    return max(1, w // 16)


def pack_codes(codes2d, lengths, wpq):
    """codes2d: (n, maxL) uint8 array of 2-bit codes (rows padded with 0).  Vectorised packer."""
    n, maxl = codes2d.shape
    width = wpq * 16
    buf = np.zeros((n, width), dtype=np.uint32)
    buf[:, :maxl] = codes2d
    sh = (2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)
    words = (buf.reshape(n, wpq, 16) << sh).sum(axis=2, dtype=np.uint64).astype(np.uint32)   # (n, wpq)
    round_up = (n + 31) // 32 * 32
    full = np.zeros((round_up, wpq), dtype=np.uint32)
    full[:n] = words
    # group of 32 reads: word w of read r at group*32*wpq + w*32 + r%32
    q = full.reshape(round_up // 32, 32, wpq).transpose(0, 2, 1).reshape(-1)
    return np.ascontiguousarray(q), np.ascontiguousarray(lengths, dtype=np.uint32)


def pack_reads(reads, max_read_length):
    n = len(reads)
    wpq = word_per_query(max_read_length)
    maxl = max([len(r) for r in reads] + [1])
    codes = np.zeros((n, maxl), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    for i, r in enumerate(reads):
        codes[i, :len(r)] = CODE[np.frombuffer(r, dtype=np.uint8)]
        lens[i] = len(r)
    q, lens = pack_codes(codes, lens, wpq)
    return q, lens, wpq


# ------------------------------------------------------------------------------------------------
# semi-global DP (s3_dp_* of the C-ABI)
# ------------------------------------------------------------------------------------------------
class DpScores(C.Structure):
    _fields_ = [("matchScore", C.c_int), ("mismatchScore", C.c_int), ("gapOpenScore", C.c_int), ("gapExtendScore", C.c_int)]


EXPORTS += ["s3_dp_config", "s3_dp_create", "s3_dp_align", "s3_dp_last_timing", "s3_dp_free"]


def pack_dp_sequences(seqs, max_len):
    """list of uint8 code arrays -> the reference DP packing: 1-based characters, 16 per word, first in the top bits,
    word w of problem t at (t//32)*(ceil(max_len/16)*32) + t%32 + 32*w   (DV-DPfunctions.cu:57-59, packers :2111-2167)"""
    n = len(seqs)
    wps = (max_len + 15) // 16
    rnd = (n + 31) // 32 * 32
    out = np.zeros(rnd * wps, dtype=np.uint32)
    for t, s in enumerate(seqs):
        L = len(s)
        buf = np.zeros(wps * 16, dtype=np.uint32)
        buf[1:L + 1] = s[:wps * 16 - 1]
        sh = (2 * (15 - np.arange(16, dtype=np.uint32))).astype(np.uint32)
        words = (buf.reshape(wps, 16) << sh).sum(axis=1, dtype=np.uint64).astype(np.uint32)
        base = (t // 32) * (wps * 32) + t % 32
        out[base + 32 * np.arange(wps)] = words
    return out


def dp_config(max_read, max_dna, want_scheme=1):
    f = lib().s3_dp_config
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    sch, tab, pat = C.c_int(), C.c_int(), C.c_int()
    _chk(f(max_read, max_dna, want_scheme, C.byref(sch), C.byref(tab), C.byref(pat)))
    return sch.value, tab.value, pat.value


class DpAligner:
    def __init__(self, max_problems, max_read, max_dna, scheme=1, scores=(1, -2, -3, -1), device=0):
        self.scheme, self.max_table, self.pattern_len = dp_config(max_read, max_dna, scheme)
        self.max_read, self.max_dna, self.max_problems = max_read, max_dna, max_problems
        self.h = C.c_void_p()
        sc = DpScores(*scores)
        f = lib().s3_dp_create
        f.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(DpScores), C.POINTER(C.c_void_p)]
        _chk(f(device, max_problems, max_read, max_dna, self.max_table, self.scheme, C.byref(sc), C.byref(self.h)))

    def close(self):
        if self.h:
            f = lib().s3_dp_free
            f.argtypes = [C.c_void_p]
            f(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def half_end(self, batch, read_lengths, occ, insert_low, insert_high, cutoffs, clip_left, clip_right):
        """default DP for half-aligned pairs: occ = structured array OCC_DTYPE of the ALIGNED ends' hits; returns (results, cigars)"""
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        occ = np.ascontiguousarray(occ, dtype=OCC_DTYPE)
        f = lib().s3_dp_half_end
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        cut = (C.c_int * 2)(*cutoffs)
        res, n, pool = C.c_void_p(), C.c_size_t(), C.c_void_p()
        t0 = time.perf_counter()
        _chk(f(self.h, batch.h, read_lengths.ctypes.data, occ.ctypes.data, len(occ), insert_low, insert_high, cut, clip_left, clip_right,
               C.byref(res), C.byref(n), C.byref(pool)))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3       # the engine itself (pack, kernels, decode, CIGAR text); what follows is this binding
        out = np.frombuffer(C.string_at(res, n.value * HALF_RESULT_DTYPE.itemsize), dtype=HALF_RESULT_DTYPE).copy() if n.value else np.zeros(0, HALF_RESULT_DTYPE)
        cigars = []
        for r in out:
            cigars.append(None if r["cigarOffset"] == 0xFFFFFFFF else C.string_at(pool.value + int(r["cigarOffset"])).decode())
        free = lib().s3_free
        free.argtypes = [C.c_void_p]
        free(res)
        free(pool)
        return out, cigars

    def deep(self, batch, seed_batch, read_lengths, pair_first_reads, stage, max_hit_num, insert_low, insert_high, cutoffs, clip_left, clip_right):
        """deep DP of pairs whose ends both failed the BWT stage; returns (results DEEP_RESULT_DTYPE, [(cigar1, cigar2)], tooManyHits flags)"""
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        ids = np.ascontiguousarray(pair_first_reads, dtype=np.uint32)
        f = lib().s3_dp_deep
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        cut = (C.c_int * 2)(*cutoffs)
        res, n, pool, tm = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_void_p()
        _chk(f(self.h, batch.h, seed_batch.h, read_lengths.ctypes.data, ids.ctypes.data, len(ids), stage, max_hit_num, insert_low, insert_high, cut, clip_left,
               clip_right, C.byref(res), C.byref(n), C.byref(pool), C.byref(tm)))
        out = np.frombuffer(C.string_at(res, n.value * DEEP_RESULT_DTYPE.itemsize), dtype=DEEP_RESULT_DTYPE).copy() if n.value else np.zeros(0, DEEP_RESULT_DTYPE)
        cigars = [(C.string_at(pool.value + int(r["cigarOffset_1"])).decode(), C.string_at(pool.value + int(r["cigarOffset_2"])).decode()) for r in out]
        too_many = np.frombuffer(C.string_at(tm, len(ids)), dtype=np.uint8).copy() if len(ids) else np.zeros(0, np.uint8)
        free = lib().s3_free
        free.argtypes = [C.c_void_p]
        for x in (res, pool, tm):
            free(x)
        return out, cigars, too_many

    def new_default(self, batch, seed_batch, read_lengths, occ, max_hit, insert_low, insert_high, cutoffs, clip_left, clip_right):
        """new default DP: occ = OCC_DTYPE hits of the aligned ends (grouped by pair, pairs ascending); their mates are seeded, both ends aligned by the DP.
        Returns (results DEEP_RESULT_DTYPE, [(cigar1, cigar2)], first reads of the pairs, flags as in deep())"""
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        occ = np.ascontiguousarray(occ, dtype=OCC_DTYPE)
        f = lib().s3_dp_new_default
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        cut = (C.c_int * 2)(*cutoffs)
        res, n, pool, fl, pf, npairs = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        _chk(f(self.h, batch.h, seed_batch.h, read_lengths.ctypes.data, occ.ctypes.data, len(occ), max_hit, insert_low, insert_high, cut, clip_left, clip_right,
               C.byref(res), C.byref(n), C.byref(pool), C.byref(fl), C.byref(pf), C.byref(npairs)))
        out = (np.frombuffer(C.string_at(res, n.value * DEEP_RESULT_DTYPE.itemsize), dtype=DEEP_RESULT_DTYPE).copy() if n.value else np.zeros(0, DEEP_RESULT_DTYPE))
        cigars = [(C.string_at(pool.value + int(r["cigarOffset_1"])).decode(), C.string_at(pool.value + int(r["cigarOffset_2"])).decode()) for r in out]
        pairs = np.frombuffer(C.string_at(pf, npairs.value * 4), dtype=np.uint32).copy() if npairs.value else np.zeros(0, np.uint32)
        flags = np.frombuffer(C.string_at(fl, npairs.value), dtype=np.uint8).copy() if npairs.value else np.zeros(0, np.uint8)
        free = lib().s3_free
        free.argtypes = [C.c_void_p]
        for x in (res, pool, fl, pf):
            free(x)
        return out, cigars, pairs, flags

    def single_end(self, batch, read_lengths, candidates, cutoff, clip_left, clip_right):
        """DP of single reads at their seeding candidates (SEED_CAND_DTYPE); returns (results SINGLE_RESULT_DTYPE, cigars)"""
        read_lengths = np.ascontiguousarray(read_lengths, dtype=np.uint32)
        cand = np.ascontiguousarray(candidates, dtype=SEED_CAND_DTYPE)
        f = lib().s3_dp_single_end
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                      C.POINTER(C.c_void_p)]
        res, n, pool = C.c_void_p(), C.c_size_t(), C.c_void_p()
        _chk(f(self.h, batch.h, read_lengths.ctypes.data, cand.ctypes.data, len(cand), cutoff, clip_left, clip_right, C.byref(res), C.byref(n), C.byref(pool)))
        out = (np.frombuffer(C.string_at(res, n.value * SINGLE_RESULT_DTYPE.itemsize), dtype=SINGLE_RESULT_DTYPE).copy() if n.value
               else np.zeros(0, SINGLE_RESULT_DTYPE))
        cigars = [C.string_at(pool.value + int(r["cigarOffset"])).decode() for r in out]
        free = lib().s3_free
        free.argtypes = [C.c_void_p]
        free(res)
        free(pool)
        return out, cigars

    def align(self, packed_dna, dna_lengths, packed_reads, read_lengths, cutoffs, clip_lt=None, clip_rt=None, anchor_lt=None, anchor_rt=None):
        n = len(read_lengths)
        u32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint32)
        packed_dna, dna_lengths, packed_reads, read_lengths = u32(packed_dna), u32(dna_lengths), u32(packed_reads), u32(read_lengths)
        clip_lt, anchor_lt, anchor_rt = u32(clip_lt), u32(anchor_lt), u32(anchor_rt)
        clip_rt = None if clip_rt is None else np.array(clip_rt, dtype=np.uint32)
        cutoffs = np.ascontiguousarray(cutoffs, dtype=np.int32)
        scores = np.zeros(n, dtype=np.int32)
        hit = np.zeros(n, dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint32)
        pat = np.zeros((n, self.pattern_len), dtype=np.uint8)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        f = lib().s3_dp_align
        f.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint32] + [C.c_void_p] * 8
        _chk(f(self.h, p(packed_dna), p(dna_lengths), p(packed_reads), p(read_lengths), p(cutoffs), n, p(clip_lt), p(clip_rt), p(anchor_lt),
               p(anchor_rt), p(scores), p(hit), p(cnt), p(pat)))
        return scores, hit, cnt, pat, clip_rt

    def timing(self):
        ms = C.c_float()
        f = lib().s3_dp_last_timing
        f.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _chk(f(self.h, C.byref(ms)))
        return ms.value


# ------------------------------------------------------------------------------------------------
# streamed single-end alignment (s3_stream_*): uploads and downloads overlap the kernels of the neighbouring batches
# ------------------------------------------------------------------------------------------------
class Stream:
    def __init__(self, index, max_reads, wpq, depth=3):
        self.h = C.c_void_p()
        self.max_reads, self.wpq, self.depth = max_reads, wpq, depth
        self.index = index
        _chk(lib().s3_stream_create(index.h, max_reads, wpq, depth, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().s3_stream_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def acquire(self):
        """numpy views of the next free slot's pinned input arrays: (queries, read lengths, seed lengths)"""
        q, l, sl = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        _chk(lib().s3_stream_acquire(self.h, C.byref(q), C.byref(l), C.byref(sl)))
        ru = (self.max_reads + 31) // 32 * 32
        return (np.ctypeslib.as_array(q, shape=(ru * self.wpq,)), np.ctypeslib.as_array(l, shape=(ru,)), np.ctypeslib.as_array(sl, shape=(ru,)))

    def submit(self, n, mm, max_out=1000, report_type=2, use_seed_lengths=False):
        t = C.c_uint64()
        _chk(lib().s3_stream_submit(self.h, n, 1 if use_seed_lengths else 0, mm, max_out, report_type, C.byref(t)))
        return t.value

    def next(self, copy=True, split=True):
        """results of the oldest submitted batch: (phase, withError, occOffsets, occPos, occStrand, occMis, deviceMs); views into pinned memory
        unless copy (they die with release()).  split=False: (..., occPos, occMeta, None, deviceMs) with occMeta = strand | mismatches << 8"""
        r = StreamResult()
        _chk(lib().s3_stream_next(self.h, C.byref(r)))
        n = r.n
        offs = np.ctypeslib.as_array(r.occOffsets, shape=(n + 1,))
        tot = int(offs[n])
        phase = np.ctypeslib.as_array(r.phase, shape=(n,))
        werr = np.ctypeslib.as_array(r.withError, shape=(n, 8))
        pos = np.ctypeslib.as_array(r.occPos, shape=(max(tot, 1),))[:tot]
        meta = np.ctypeslib.as_array(r.occMeta, shape=(max(tot, 1),))[:tot]
        if copy:
            phase, werr, offs, pos, meta = phase.copy(), werr.copy(), offs.copy(), pos.copy(), meta.copy()
        if not split:
            return phase, werr, offs, pos, meta, None, r.deviceMs
        return phase, werr, offs, pos, (meta & 0xFF).astype(np.uint8), ((meta >> 8) & 0xFF).astype(np.uint8), r.deviceMs

    def release(self):
        _chk(lib().s3_stream_release(self.h))


def stream_bench(index, queries, lens, wpq, mm, max_out, batch_reads, steps, depth=3, check=None):
    """`steps` passes over the reads in batches of batch_reads through a Stream: the packed reads go host -> pinned -> device and the occurrence
    lists device -> pinned inside the timed region.  check(first, phase, offs, pos, strand, mis) is called per batch of the first pass (outside the
    clock is not possible here: it runs between batches, keep it cheap).  Returns {"reads", "seconds", "occurrences", "batch_reads", "depth"}."""
    import time
    n = len(lens)
    batch_reads = min(batch_reads, n) // 32 * 32
    st = Stream(index, batch_reads, wpq, depth)
    starts = list(range(0, n - n % 32, batch_reads))
    jobs = [(a, min(batch_reads, n - n % 32 - a)) for a in starts]
    occ = 0
    try:
        # one untimed pass warms the slots (buffers grow to their working size)
        for timed in (False, True):
            reps = steps if timed else 1
            pending = []
            t0 = time.perf_counter()
            for rep in range(reps):
                for (a, m) in jobs:
                    if len(pending) == depth:
                        ph, we, of, po, me, _, ms = st.next(copy=False, split=False)
                        if timed:
                            occ += int(of[-1])
                        elif check is not None:
                            check(pending[0][0], ph, of, po, (me & 0xFF).astype(np.uint8), ((me >> 8) & 0xFF).astype(np.uint8))
                        st.release()
                        pending.pop(0)
                    q, l, _ = st.acquire()
                    ru = (m + 31) // 32 * 32
                    q[:ru * wpq] = queries[a * wpq:(a + ru) * wpq]
                    l[:m] = lens[a:a + m]
                    st.submit(m, mm, max_out, 2)
                    pending.append((a, m))
            while pending:
                ph, we, of, po, me, _, ms = st.next(copy=False, split=False)
                if timed:
                    occ += int(of[-1])
                elif check is not None:
                    check(pending[0][0], ph, of, po, (me & 0xFF).astype(np.uint8), ((me >> 8) & 0xFF).astype(np.uint8))
                st.release()
                pending.pop(0)
            dt = time.perf_counter() - t0
    finally:
        st.close()
    return {"reads": sum(m for _, m in jobs) * steps, "seconds": dt, "occurrences": occ // max(1, steps), "batch_reads": batch_reads, "depth": depth,
            "what": "packed reads host -> device and occurrence lists device -> host inside the timed region, %d batches in flight" % depth}


# ------------------------------------------------------------------------------------------------
# several devices (s3_group_*): index replicated per member, batches dealt in turn, one RCCL all-reduce of the counters
# ------------------------------------------------------------------------------------------------
class _MemberIndex:
    """an index owned by a Group (not freed on its own)"""

    def __init__(self, h):
        self.h = C.c_void_p(h)

    def device_bytes(self):
        return lib().s3_index_device_bytes(self.h)

    def close(self):
        pass


class Group:
    def __init__(self, host_index, devices):
        self.h = C.c_void_p()
        d = host_index.desc()
        arr = (C.c_int * len(devices))(*devices)
        _chk(lib().s3_group_create(arr, len(devices), C.byref(d), C.byref(self.h)))
        self.n = len(devices)

    def index(self, k):
        return _MemberIndex(lib().s3_group_index(self.h, k))

    def member_of_batch(self, i):
        return lib().s3_group_member_of_batch(self.h, i)

    def allreduce(self, rows):
        """rows: (members, counters) array of uint64; returns the array with every row = column sums"""
        a = np.ascontiguousarray(rows, dtype=np.uint64).copy()
        assert a.shape[0] == self.n
        _chk(lib().s3_group_allreduce(self.h, a.ctypes.data, a.shape[1]))
        return a

    def close(self):
        if self.h:
            lib().s3_group_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
