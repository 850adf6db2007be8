"""2BWT / GPU-2BWT index construction from a 2-bit text, on the GPU (torch is used only for device
memory, sort, scan and gather -- plumbing).

Produces exactly the payloads the reference tools write (soap3-dp-builder + BGS-Build; formats in
SURVEY.md 5.9): .bwt/.rev.bwt code words, .fmv.gpu/.rev.fmv.gpu sampled ranks, the full suffix array,
the word-packed text and both 13-mer lookup tables.  tests/test_builder.py checks byte equality with
index files written by the reference builder.

Suffix sorting: one radix sort of 31-mer keys per 2-character bucket, then prefix doubling restricted
to the groups that are still tied (repeats longer than 31 characters).  Suffixes that run off the end
of the text compare as if followed by '$' < A.
"""
import sys

import numpy as np
import torch

K0 = 31  # characters in the first-round key (62 bits, stays positive in int64)


def _kmer_keys(codes, k, chunk=1 << 28):
    """keys[p] = 2-bit packing of codes[p:p+k], zero padded beyond the end (int64).  Built from k shifted
    contiguous slices per chunk of positions, so no gathers and only one n-sized output."""
    n = codes.numel()
    dev = codes.device
    padded = torch.zeros(n + k, dtype=torch.uint8, device=dev)
    padded[:n] = codes
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        acc = torch.zeros(b - a, dtype=torch.int64, device=dev)
        for j in range(k):
            acc = (acc << 2) | padded[a + j:b + j].to(torch.int64)
        keys[a:b] = acc
    return keys


CH = 1 << 30   # every elementwise / scan / nonzero call is issued on at most this many elements (32-bit safe)


def _nonzero_big(mask):
    out = []
    n = mask.numel()
    for a in range(0, n, CH):
        nz = torch.nonzero(mask[a:min(n, a + CH)]).squeeze(1)
        if a:
            nz = nz + a
        out.append(nz)
    return out[0] if len(out) == 1 else torch.cat(out)


def _cumsum_big(x):
    """inclusive cumsum of a bool/int tensor as int64, chunked with carry"""
    n = x.numel()
    out = torch.empty(n, dtype=torch.int64, device=x.device)
    carry = 0
    for a in range(0, n, CH):
        b = min(n, a + CH)
        c = torch.cumsum(x[a:b].to(torch.int64), 0)
        if carry:
            c += carry
        out[a:b] = c
        carry = int(c[-1])
    return out


def _group_starts(newgrp, slots=None):
    """for a boolean 'new group starts here' vector: the slot value of the group's first member, per element.
    slots: the values to propagate (default: the element's own index)"""
    n = newgrp.numel()
    dev = newgrp.device
    out = torch.empty(n, dtype=torch.int64, device=dev)
    carry = 0
    for a in range(0, n, CH):
        b = min(n, a + CH)
        v = slots[a:b] if slots is not None else torch.arange(a, b, dtype=torch.int64, device=dev)
        v = torch.where(newgrp[a:b], v, torch.zeros_like(v))
        v = torch.cummax(v, 0).values
        if carry:
            v = torch.clamp(v, min=carry)
        out[a:b] = v
        carry = int(v[-1])
    return out


def _gather_big(src, idx):
    n = idx.numel()
    out = torch.empty(n, dtype=src.dtype, device=src.device)
    for a in range(0, n, CH):
        b = min(n, a + CH)
        out[a:b] = src[idx[a:b]]
    return out


def _scatter_big(dst, idx, val):
    n = idx.numel()
    for a in range(0, n, CH):
        b = min(n, a + CH)
        dst[idx[a:b]] = val[a:b]


def _tied(newgrp):
    """members of groups of size > 1"""
    n = newgrp.numel()
    t = torch.empty(n, dtype=torch.bool, device=newgrp.device)
    t[:-1] = ~(newgrp[:-1] & newgrp[1:])
    t[-1] = ~newgrp[-1]
    return t


def suffix_array(codes, max_bucket=1 << 30, verbose=False):
    """codes: uint8 tensor (values 0..3).  Returns int64 tensor sa of length n: sorted suffix start positions
    (the '$' suffix, SA[0] = n in the reference's numbering, is NOT included)."""
    n = codes.numel()
    dev = codes.device
    keys = _kmer_keys(codes, K0)
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    newgrp = torch.ones(n, dtype=torch.bool, device=dev)
    # bucket by leading characters so that every sort is well below 2^31 elements
    lead_chars = 0
    while (n >> (2 * lead_chars)) > max_bucket // 2 and lead_chars < 8:
        lead_chars += 1
    nb = 1 << (2 * lead_chars)
    if nb == 1:
        ks, order = torch.sort(keys)
        sa[:] = order
        newgrp[1:] = ks[1:] != ks[:-1]
        del ks, order
    else:
        shift = 2 * (K0 - lead_chars)
        off = 0
        for b in range(nb):
            parts = []
            for a in range(0, n, CH):
                e = min(n, a + CH)
                nz = torch.nonzero((keys[a:e] >> shift) == b).squeeze(1)
                parts.append(nz + a if a else nz)
            pos = parts[0] if len(parts) == 1 else torch.cat(parts)
            del parts
            m = pos.numel()
            if m == 0:
                continue
            ks, o = torch.sort(keys[pos])
            sa[off:off + m] = pos[o]
            newgrp[off + 1:off + m] = ks[1:] != ks[:-1]          # a bucket boundary is always a group boundary
            off += m
            del pos, ks, o
    del keys
    unresolved = _tied(newgrp)
    if not bool(unresolved.any()):
        if verbose:
            print("  [sa] all suffixes distinct within %d characters" % K0, file=sys.stderr)
        return sa
    # rank[p] = SA index of the first member of p's group
    gstart = _group_starts(newgrp)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    _scatter_big(rank, sa, gstart)
    del gstart, newgrp
    h = K0
    rounds = 0
    while True:
        sidx = _nonzero_big(unresolved)                      # SA slots still tied, ascending
        m = sidx.numel()
        if verbose:
            print("  [sa] h=%d unresolved=%d" % (h, m), file=sys.stderr)
        if m == 0:
            break
        pos = _gather_big(sa, sidx)
        g = _gather_big(rank, pos)                           # group id = SA index of group start
        q = pos + h
        r2 = torch.where(q < n, _gather_big(rank, torch.clamp(q, max=n - 1)), n - q - 1)   # beyond the end: shorter suffix first
        # dense group labels keep the composite key inside int64
        gl = torch.ones(m, dtype=torch.bool, device=dev)
        gl[1:] = g[1:] != g[:-1]
        dense = _cumsum_big(gl) - 1
        span = n + h + 2
        comp = dense * span + (r2 + h + 1)
        del g, q, r2, gl, dense
        if m >= (1 << 31):
            raise RuntimeError("too many tied suffixes for one sort (%d); text is too repetitive for this builder" % m)
        cs, o = torch.sort(comp)
        del comp
        pos_sorted = pos[o]
        del pos, o
        _scatter_big(sa, sidx, pos_sorted)
        newg = torch.ones(m, dtype=torch.bool, device=dev)
        newg[1:] = cs[1:] != cs[:-1]
        del cs
        gs = _group_starts(newg, sidx)
        _scatter_big(rank, pos_sorted, gs)
        del gs, pos_sorted
        unresolved[sidx] = _tied(newg)
        del newg, sidx
        h *= 2
        rounds += 1
        if rounds > 40:
            raise RuntimeError("suffix sort did not converge")
    return sa


def _pack_msb_words(codes, n_words):
    """codes (uint8 tensor) -> words (int64 holding uint32 values), 16 chars per word, first char in the top bits"""
    dev = codes.device
    n = codes.numel()
    out = torch.empty(n_words, dtype=torch.int64, device=dev)
    sh = (2 * (15 - torch.arange(16, dtype=torch.int64, device=dev)))
    wch = CH // 16
    for a in range(0, n_words, wch):
        b = min(n_words, a + wch)
        buf = torch.zeros((b - a) * 16, dtype=torch.int64, device=dev)
        lo, hi = a * 16, min(n, b * 16)
        if hi > lo:
            buf[:hi - lo] = codes[lo:hi].to(torch.int64)
        out[a:b] = (buf.view(b - a, 16) << sh).sum(dim=1)
    return out


def _half_index(codes, verbose=False):
    """BWT code words, GPU-2BWT occ table, inverseSa0, cumulative frequencies and SA for one text"""
    n = codes.numel()
    dev = codes.device
    sa = suffix_array(codes, verbose=verbose)
    # full SA including the '$' suffix at index 0
    sa_full = torch.empty(n + 1, dtype=torch.int64, device=dev)
    sa_full[0] = n
    sa_full[1:] = sa
    del sa
    inv0 = -1
    for a in range(0, n + 1, CH):
        z = torch.nonzero(sa_full[a:min(n + 1, a + CH)] == 0)
        if z.numel():
            inv0 = a + int(z[0])
    # stored BWT: text[SA[i]-1] for every i except the one where SA[i] = 0 ('$' is not stored; BWT.c:169-194)
    stored = torch.empty(n, dtype=torch.uint8, device=dev)
    for a in range(0, n + 1, CH):
        b = min(n + 1, a + CH)
        ch = codes[torch.clamp(sa_full[a:b] - 1, min=0)]
        # destination index: i for i < inv0, i-1 for i > inv0
        if b <= inv0:
            stored[a:b] = ch
        elif a > inv0:
            stored[a - 1:b - 1] = ch
        else:
            k = inv0 - a
            stored[a:inv0] = ch[:k]
            stored[inv0:b - 1] = ch[k + 1:]
    cnt = [0, 0, 0, 0]
    for a in range(0, n, CH):
        seg = codes[a:min(n, a + CH)]
        for c in range(4):
            cnt[c] += int((seg == c).sum())
    cum = [0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2], n]
    n_words = (n + 15) // 16
    bwt_words = _pack_msb_words(stored, n_words)
    # sampled ranks every 128 characters over the zero-padded stored string, + C[c]  (BGS-Build.cpp:139-161)
    recs = (n + 127) // 128 + 1
    per = torch.zeros(((recs - 1), 4), dtype=torch.int64, device=dev)
    bch = CH // 128
    for a in range(0, recs - 1, bch):
        b = min(recs - 1, a + bch)
        seg = torch.zeros((b - a) * 128, dtype=torch.uint8, device=dev)
        lo, hi = a * 128, min(n, b * 128)
        if hi > lo:
            seg[:hi - lo] = stored[lo:hi]
        seg = seg.view(b - a, 128)
        for c in range(4):
            per[a:b, c] = (seg == c).sum(dim=1)
    occ = torch.zeros((recs, 4), dtype=torch.int64, device=dev)
    occ[1:] = torch.cumsum(per, 0)
    occ += torch.tensor(cum[:4], dtype=torch.int64, device=dev)
    return dict(inverseSa0=inv0, cum=cum, bwt=bwt_words, occ=occ.view(-1), sa=sa_full)


def _lookup_table(codes, k=13):
    n = codes.numel()
    dev = codes.device
    padded = torch.zeros(n + k, dtype=torch.uint8, device=dev)
    padded[:n] = codes
    cnt = torch.zeros(1 << (2 * k), dtype=torch.int64, device=dev)
    step = min(CH, 1 << 26)          # torch.bincount on ROCm dies (SIGFPE) on 2^30-element inputs: keep calls small
    for a in range(0, n, step):
        b = min(n, a + step)
        acc = torch.zeros(b - a, dtype=torch.int64, device=dev)
        for j in range(k):
            acc = (acc << 2) | padded[a + j:b + j].to(torch.int64)
        cnt += torch.bincount(acc, minlength=1 << (2 * k))
    return torch.cumsum(cnt, 0)


def _u32(t):
    return (t.to(torch.int64) & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


def build_host_index(codes_np, device="cpu", verbose=False):
    """codes_np: uint8 numpy array of 2-bit codes (N-runs already removed).  Returns an s3dp.HostIndex."""
    from . import HostIndex, pack_text_words
    dev = torch.device(device)
    codes = torch.from_numpy(np.ascontiguousarray(codes_np)).to(dev)
    hi = HostIndex()
    hi.n = int(codes.numel())
    f = _half_index(codes, verbose)
    hi.sa = _u32(f["sa"])
    hi.sa[0] = 0xFFFFFFFF
    hi.fwd = dict(inverseSa0=f["inverseSa0"], cum=f["cum"], bwt=_u32(f["bwt"]), occ=_u32(f["occ"]))
    del f
    rcodes = torch.flip(codes, [0])
    r = _half_index(rcodes, verbose)
    hi.rev = dict(inverseSa0=r["inverseSa0"], cum=r["cum"], bwt=_u32(r["bwt"]), occ=_u32(r["occ"]))
    del r
    hi.lkt = _u32(_lookup_table(codes))
    hi.rev_lkt = _u32(_lookup_table(rcodes))
    hi.lkt_size = 13
    hi.codes = codes_np
    n = hi.n
    words = (n + 63) // 64 * 4 + 8
    hi.pac = _u32(_pack_msb_words(codes, words))
    return hi
