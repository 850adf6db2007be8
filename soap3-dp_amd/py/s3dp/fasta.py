"""FASTA front end of the index builder: what the reference's `soap3-dp-builder` does before it sorts suffixes (HSPParseFASTAToPacked,
2bwt-lib/HSP.c:677-1086, called from 2bwt-flex/2BWT-Builder.c:333) -- restated, vectorised with numpy so that a 3 Gbp genome parses in seconds.

Behaviour that decides text positions, and so every SAM coordinate (kept exactly):
  * a record starts at '>' (the file must begin with one); its name is the header line up to the first white-space character, at most 256
    characters; a name `gi|<n>|...` carries a GI number (HSP.c:781-784);
  * the sequence runs to the next '>' -- ANYWHERE, not only at a line start -- or to the end of the file; '\\n' and '\\t' are skipped, every other
    byte is a character: A C G T in either case are kept, anything else (N, IUPAC codes, '\\r', blanks ...) is ambiguous (HSP.c:855-895);
  * a run of fewer than 10 ambiguous characters becomes that many 'G'; a run of 10 or more is REMOVED from the text and recorded; a run ends at an
    unambiguous character, at '>' or at the end of the file (line breaks inside a run do not end it);
  * `-U` (MaskLowerCase): a lower-case letter OPENS a run of ambiguous characters; only real ambiguity codes extend one (HSP.c:849-870), so a
    soft-masked stretch becomes 'G's, base by base, and is not removed.
The removed runs and the sequence boundaries go into the translate table (.tra) that turns text positions back into (sequence, offset): entries
sorted by start, a 2^18-sampled grid of how many entries start before each grid cell, and the sequences' real offsets -- including the reference's
way of computing a run's length from the CUMULATIVE count of discarded characters (HSP.c:886-887, :959): file formats are what other tools read.
"""
import numpy as np

GRID_SAMPLING_FACTOR = 262144                     # 2bwt-lib/HSP.h:70
MAX_SEQ_NAME_LENGTH = 256                         # 2bwt-lib/HSP.h:242
_SPACE = frozenset(b" \t\n\v\f\r")

_CODE = np.full(256, 255, dtype=np.uint8)         # 0..3: A C G T (either case); 255: ambiguous
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c + 32] = _i
_CODE_MASKED = _CODE.copy()
_CODE_MASKED[ord("a"):ord("z") + 1] = 255         # -U: lower case counts as 'N'


class ParsedFasta:
    """codes: uint8 2-bit codes of the indexed text; seqs: [(gi, name, start, length)] in text coordinates; amb: [(start, rightOfEnd)] of the removed
    runs as the reference records them; actual: [(start, length)] of every sequence in FASTA coordinates (removed characters included)."""

    def __init__(self):
        self.codes = np.zeros(0, dtype=np.uint8)
        self.seqs, self.amb, self.actual = [], [], []
        self.seed = 0

    @property
    def n(self):
        return int(self.codes.size)


def parse_fasta(src, seed=0, mask_lower_case=False):
    """src: a path or the bytes of a FASTA file"""
    data = np.fromfile(src, dtype=np.uint8) if isinstance(src, str) else np.frombuffer(bytes(src), dtype=np.uint8)
    if data.size == 0 or data[0] != ord(">"):
        raise ValueError("FASTA file does not begin with '>'")
    lut = _CODE_MASKED if mask_lower_case else _CODE
    gt = np.flatnonzero(data == ord(">"))
    nl = np.flatnonzero(data == ord("\n"))
    out = ParsedFasta()
    out.seed = seed
    pieces = []
    total = 0                                     # characters of the text so far
    discarded = 0                                 # removed characters so far (all sequences)
    pos = 0                                       # at a '>'
    end = int(data.size)
    while pos < end:
        k = int(np.searchsorted(nl, pos))         # the header line ends at the next '\n' (a '>' inside it is part of the name)
        hdr_end = int(nl[k]) if k < nl.size else end
        header = bytes(data[pos + 1:hdr_end])
        name = bytearray()
        for c in header:
            if c in _SPACE:
                break
            if len(name) < MAX_SEQ_NAME_LENGTH:
                name.append(c)
        name = name.decode("latin-1")
        gi = 0
        if len(name) > 3 and name.startswith("gi|"):
            digits = ""
            for ch in name[3:]:
                if not ch.isdigit():
                    break
                digits += ch
            gi = int(digits) & 0xFFFFFFFF if digits else 0
        body = hdr_end + 1
        j = int(np.searchsorted(gt, body))
        nxt = int(gt[j]) if j < gt.size else end  # the sequence runs to the next '>' wherever it stands
        chunk = data[body:nxt] if body < nxt else data[0:0]
        chunk = chunk[(chunk != ord("\n")) & (chunk != ord("\t"))]
        code = lut[chunk]
        ambiguous = code == 255
        seq_start, actual_start = total, total + discarded
        if ambiguous.any():
            if mask_lower_case:
                # -U as the reference applies it (HSP.c:849-870): the mask turns a lower-case letter into an ambiguous character where a run may START
                # (the outer loop); the loop that counts the run tests the UNMASKED map, where a / c / g / t are ordinary bases and end it.  So a masked
                # base opens a run of its own -- one 'G' unless real ambiguity codes follow it -- and only N / IUPAC characters (either case) extend a run.
                extends = _CODE[chunk] == 255
                prev = np.concatenate(([False], ambiguous[:-1]))
                start = ambiguous & (~extends | ~prev)
                rs = np.flatnonzero(start)
                brk = np.flatnonzero(start | ~ambiguous)                        # where a run cannot continue into
                nxt_brk = np.searchsorted(brk, rs, side="right")
                re_ = np.where(nxt_brk < brk.size, brk[np.minimum(nxt_brk, brk.size - 1)], chunk.size)
            else:
                d = np.diff(np.concatenate(([0], ambiguous.view(np.int8), [0])))
                rs, re_ = np.flatnonzero(d == 1), np.flatnonzero(d == -1)      # runs [rs, re_)
            rlen = re_ - rs
            remove = rlen >= 10
            keep = np.ones(chunk.size, dtype=bool)
            # characters kept in front of every run: its start in the sequence minus what earlier runs of this sequence lost
            lost_before = np.concatenate(([0], np.cumsum(np.where(remove, rlen, 0))[:-1]))
            for s0, e0, ln, rm, lb in zip(rs.tolist(), re_.tolist(), rlen.tolist(), remove.tolist(), lost_before.tolist()):
                if rm:
                    keep[s0:e0] = False
                    discarded += ln
                    start = seq_start + s0 - lb
                    out.amb.append((start, start + discarded - 1))               # (the reference's rightOfEndPos: cumulative, HSP.c:886-887)
            code = np.where(ambiguous, np.uint8(2), code)[keep]                 # short runs: 'G'
        num = int(code.size)
        pieces.append(code.astype(np.uint8, copy=False))
        out.seqs.append((gi, name, seq_start, num))
        total += num
        out.actual.append((actual_start, total + discarded - actual_start))
        pos = nxt
    out.codes = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.uint8)
    return out


def translate_table(p):
    """the entries of the .tra file: (grid, [(startPos, chrID, correction)]) -- HSP.c:940-1035, 32-bit wrap-around included"""
    n, num_amb, num_seq = p.n, len(p.amb), len(p.seqs)
    grid_entries = n // GRID_SAMPLING_FACTOR + 1
    grid = np.zeros(grid_entries, dtype=np.int64)
    tr = []
    M = 0xFFFFFFFF

    def amb_entry(j, k):
        start, right = p.amb[j]
        blk = right - start + 1
        base = p.actual[k - 1][0] if k > 0 else 0
        grid[start // GRID_SAMPLING_FACTOR] += 1
        tr.append((start, k, (base - blk - 1) & M))

    def seq_entry(k):
        start = p.seqs[k][2]
        grid[start // GRID_SAMPLING_FACTOR] += 1
        tr.append((start, k + 1, (start - 1) & M))

    j = k = 0
    while j < num_amb and k < num_seq:
        a, s = p.amb[j][0], p.seqs[k][2]
        if a < s:
            amb_entry(j, k)
            j += 1
        elif a > s:
            seq_entry(k)
            k += 1
        else:
            k += 1                                # a removed run right at a sequence start: no entry for the boundary
    while j < num_amb:
        amb_entry(j, k)
        j += 1
    while k < num_seq:
        seq_entry(k)
        k += 1
    while len(tr) < num_amb + num_seq:            # (entries skipped above are filled with copies of the last one)
        tr.append(tr[-1] if tr else (0, 0, 0))
    grid = np.cumsum(grid) - 1
    return [int(x) & M for x in grid], tr


def pac_bytes(codes):
    """.pac: four characters per byte, the first in the top bits; then one (partial or empty) byte's worth and the count n % 4 (HSP.c:926-938)"""
    n = int(codes.size)
    pad = (-n) % 4
    c = np.concatenate((codes, np.zeros(pad, dtype=np.uint8))).reshape(-1, 4)
    packed = (c[:, 0] << 6) | (c[:, 1] << 4) | (c[:, 2] << 2) | c[:, 3]
    tail = bytes([0, 0]) if n % 4 == 0 else bytes([n % 4])
    return packed.astype(np.uint8).tobytes() + tail


def write_annotation_files(p, prefix):
    """<prefix>.ann .amb .tra as the reference builder writes them (HSP.c:1037-1080)"""
    n, num_seq = p.n, len(p.seqs)
    with open(prefix + ".ann", "w", encoding="latin-1") as f:
        f.write("%u %u %u\n" % (n, num_seq, p.seed))
        for gi, name, start, length in p.seqs:
            f.write("%u %s\n%u %u 0\n" % (gi, name, start, length))
    with open(prefix + ".amb", "w") as f:
        f.write("%u %u 0\n" % (n, num_seq))
    grid, tr = translate_table(p)
    with open(prefix + ".tra", "w") as f:
        f.write("%u %u %u %u\n" % (n, num_seq, len(p.amb), len(grid)))
        f.write("".join("%u\n" % g for g in grid))
        f.write("".join("%u %u %u\n" % t for t in tr))
        f.write("".join("%u %u\n" % a for a in p.actual))


def build_index_files(fasta, prefix, device="cpu", seed=0, mask_lower_case=False, with_lkt=True, verbose=False):
    """FASTA -> every file `soap3-dp` loads (<prefix>.bwt .fmv.gpu .rev.bwt .rev.fmv.gpu .sa .pac .ann .amb .tra .lkt .rev.lkt): the front end above,
    then the suffix sort and the tables of s3dp.builder on `device` ("cuda" on a GPU box).  Replaces `soap3-dp-builder ref.fa` + `BGS-Build`."""
    from . import builder
    p = parse_fasta(fasta, seed=seed, mask_lower_case=mask_lower_case)
    if p.n < 16:
        raise ValueError("the FASTA file holds fewer than 16 usable characters")
    hi = builder.build_host_index(p.codes, device=device, verbose=verbose)
    hi.write_files(prefix, with_lkt=with_lkt)       # (writes single-sequence .ann/.amb/.tra, replaced below)
    write_annotation_files(p, prefix)
    return p, hi


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m s3dp.fasta", description="build the 2BWT + GPU-2BWT index files of a FASTA file (soap3-dp-builder + BGS-Build)")
    ap.add_argument("fasta")
    ap.add_argument("-o", "--prefix", default=None, help="index prefix (default: <fasta>.index, as the reference)")
    ap.add_argument("--device", default=None, help="cpu / cuda (default: cuda when a GPU is visible)")
    ap.add_argument("-U", dest="mask", action="store_true", help="treat lower-case letters as N")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    dev = a.device
    if dev is None:
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
    p, _ = build_index_files(a.fasta, a.prefix or a.fasta + ".index", device=dev, seed=a.seed, mask_lower_case=a.mask, verbose=True)
    print("[builder] %d sequences, %d characters indexed, %d ambiguous runs removed" % (len(p.seqs), p.n, len(p.amb)))


if __name__ == "__main__":
    main()
