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


def suffix_array(codes, max_bucket=1 << 30, verbose=False):
    """codes: uint8 tensor (values 0..3).  Returns int64 tensor sa of length n: sorted suffix start positions
    (the '$' suffix, SA[0] = n in the reference's numbering, is NOT included)."""
    n = codes.numel()
    dev = codes.device
    keys = _kmer_keys(codes, K0)
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    # bucket by leading characters so that every sort is well below 2^31 elements
    lead_chars = 0
    while (n >> (2 * lead_chars)) > max_bucket // 2 and lead_chars < 8:
        lead_chars += 1
    nb = 1 << (2 * lead_chars)
    if nb == 1:
        order = torch.argsort(keys)
        sa[:] = order
        skeys = keys[order]
    else:
        bid = (keys >> (2 * (K0 - lead_chars))).to(torch.int16)
        skeys = torch.empty(n, dtype=torch.int64, device=dev)
        off = 0
        for b in range(nb):
            pos = torch.nonzero(bid == b).squeeze(1)
            if pos.numel() == 0:
                continue
            kb = keys[pos]
            ks, o = torch.sort(kb)
            m = pos.numel()
            sa[off:off + m] = pos[o]
            skeys[off:off + m] = ks
            off += m
            del pos, kb, ks, o
        del bid
    del keys
    # group starts: rank[p] = SA index of the first member of p's group
    newgrp = torch.ones(n, dtype=torch.bool, device=dev)
    newgrp[1:] = skeys[1:] != skeys[:-1]
    del skeys
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    gstart = torch.cummax(torch.where(newgrp, idx, torch.zeros_like(idx)), 0).values
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[sa] = gstart
    # unresolved = members of groups of size > 1
    nxt = torch.ones(n, dtype=torch.bool, device=dev)
    nxt[:-1] = newgrp[1:]
    unresolved = ~(newgrp & nxt)
    del nxt, newgrp, gstart
    h = K0
    rounds = 0
    while True:
        sidx = torch.nonzero(unresolved).squeeze(1)          # SA slots still tied, ascending
        m = sidx.numel()
        if verbose:
            print("  [sa] h=%d unresolved=%d" % (h, m))
        if m == 0:
            break
        pos = sa[sidx]
        g = rank[pos]                                        # group id = SA index of group start
        q = pos + h
        r2 = torch.where(q < n, rank[torch.clamp(q, max=n - 1)], n - q - 1)    # beyond the end: shorter suffix first
        # dense group labels keep the composite key inside int64
        gl = torch.ones(m, dtype=torch.bool, device=dev)
        gl[1:] = g[1:] != g[:-1]
        dense = torch.cumsum(gl.to(torch.int64), 0) - 1
        span = n + h + 2
        comp = dense * span + (r2 + h + 1)
        cs, o = torch.sort(comp)
        pos_sorted = pos[o]
        sa[sidx] = pos_sorted
        newg = torch.ones(m, dtype=torch.bool, device=dev)
        newg[1:] = cs[1:] != cs[:-1]
        gs = torch.cummax(torch.where(newg, sidx, torch.zeros_like(sidx)), 0).values
        rank[pos_sorted] = gs
        nx = torch.ones(m, dtype=torch.bool, device=dev)
        nx[:-1] = newg[1:]
        still = ~(newg & nx)
        unresolved[sidx] = still
        h *= 2
        rounds += 1
        if rounds > 40:
            raise RuntimeError("suffix sort did not converge")
    return sa


def _pack_msb_words(codes, n_words):
    """codes (uint8 tensor) -> uint32 words, 16 chars per word, first char in the most significant bits"""
    dev = codes.device
    buf = torch.zeros(n_words * 16, dtype=torch.int64, device=dev)
    buf[:codes.numel()] = codes.to(torch.int64)
    sh = (2 * (15 - torch.arange(16, dtype=torch.int64, device=dev)))
    w = (buf.view(n_words, 16) << sh).sum(dim=1)
    return w


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
    inv0 = int(torch.nonzero(sa_full == 0).item())
    prev = sa_full - 1
    keep = torch.ones(n + 1, dtype=torch.bool, device=dev)
    keep[inv0] = False
    stored = codes[prev[keep]]                                # n characters, '$' dropped (BWT.c:169-194)
    del prev, keep
    cnt = torch.bincount(codes.to(torch.int64), minlength=4)
    cum = [0] + [int(x) for x in torch.cumsum(cnt, 0)]
    n_words = (n + 15) // 16
    bwt_words = _pack_msb_words(stored, n_words)
    # sampled ranks every 128 characters over the zero-padded stored string, + C[c]  (BGS-Build.cpp:139-161)
    recs = (n + 127) // 128 + 1
    padded = torch.zeros((recs - 1) * 128, dtype=torch.int64, device=dev)
    padded[:n] = stored.to(torch.int64)
    blk = torch.arange((recs - 1) * 128, dtype=torch.int64, device=dev) // 128
    per = torch.bincount(blk * 4 + padded, minlength=(recs - 1) * 4).view(recs - 1, 4)
    occ = torch.zeros((recs, 4), dtype=torch.int64, device=dev)
    occ[1:] = torch.cumsum(per, 0)
    occ += torch.tensor(cum[:4], dtype=torch.int64, device=dev)
    return dict(inverseSa0=inv0, cum=cum, bwt=bwt_words, occ=occ.view(-1), sa=sa_full)


def _lookup_table(codes, k=13):
    n = codes.numel()
    keys = _kmer_keys(codes, k)
    cnt = torch.bincount(keys, minlength=1 << (2 * k))
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
