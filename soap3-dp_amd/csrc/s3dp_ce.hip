// s3dp_ce.hip -- check-and-extend of the 2BWT k-mismatch search for gfx950 (BGS-HostAlgnmtAlgo2.cpp:36-176): the stepping kernels (s3dp_first.hip: first
// pass and trunk pass; s3dp_unit.hip: unit pass) do not walk a small SA range inside a model's check window: they write a TASK record (range, where the
// search stands in the model) into the record stream, reserve one sequence number per candidate and go on.  This kernel verifies the candidates -- suffix-
// array value, then the text under the rest of the model against the oriented read, 16 characters per XOR / fold / popcount -- and appends the accepted
// ones under their reserved numbers, so the record stream keeps the CPU path's order; rejected candidates leave holes the ordering pass skips.
// Since round 6 a candidate is first looked up in a FLANK TABLE (s3dp_flank.hip): one 16-byte record {text position, the 48 characters next to the matched
// part} in the order of the index the handing-over step walks -- the forward index for backward steps (recs[].z), the reverse index for forward steps
// (SearchOut::recAux) -- and nine in ten are rejected from that record alone; survivors go on as before.  A forward task's candidates come in the REVERSE
// index's order: a survivor finds its place in the forward range (its sequence number) by looking its text position up in the task's <= 32 suffix-array entries.
// (Round 5 tried the text a second time in OVERLAPPING 64-byte lines -- line k = the 256 characters from 128 k on, so that no window straddles two lines:
//  1.0 text lines per candidate instead of 1.4 -- and the kernel got SLOWER, 72.4 -> 78.2 ms per ladder: twice the text's bytes halve what the 256 MB
//  of last-level cache hold of it, and that costs more than the second lines did.  profiles/README_r05.md.)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "s3dp_device.h"
#include "s3dp_model.h"
#include "s3dp_search_common.h"

namespace s3 {

using namespace v2;

// ---- check-and-extend of the task records --------------------------------------------------------------------------------------------------
// A wave takes 64 records at a time; the candidates of their tasks are numbered through (wave prefix sum of the task sizes, descriptors in LDS) and
// dealt to the lanes 64 at a time -- one lane per CANDIDATE, whatever the sizes of the tasks (1 .. 256): a lane finds its task by bisection in the
// prefix sums.  TWO PHASES since round 6:
//   FILTER   every candidate: its flank record (one 16-byte load, no dependent second one) against the read's characters next to the matched part.  Nine in ten
//            end here.  Survivors -- and the candidates of tasks no table serves -- go into a queue in LDS with all that the second phase needs (seven words),
//            so that the queue outlives the batch of records it came from;
//   VERIFY   whenever the queue holds 64 survivors (and once more at the end): suffix-array value (from the flank record; a forward task's survivor finds its
//            place in the FORWARD range -- its sequence number -- among the task's suffix-array entries), then per remaining model step the text against the
//            oriented read, 16 characters per XOR / fold / popcount, the step's error window checked as in the stepping kernel (BGS-HostAlgnmtAlgo2.cpp:86-169).
//            Accepted candidates are appended to the record stream with one atomic per wave and S3_V2_REC_BLOCK records.
// The first form of the round-6 kernel filtered and verified in one round of 64 candidates: a round then waited for its few survivors' two or three dependent
// loads, and the kernel got SLOWER than without the tables (64.6 -> 82.8 ms per ladder); the verification has to run on full waves.
// COUNT (counters enabled: measurement passes only): besides the tallies, WHERE the rejected candidates are decided.  Per rejected candidate: the direction
// of the step it was handed over in (backward: the characters to its LEFT; forward: the characters behind the matched part), the mismatches it may still take
// in that step (0 / 1 / 2 and more), and whether its flank record alone rejected it (without a table: whether the 32 characters next to the matched part
// alone would have -- the measurement the tables were built on, 91.7 % on the benchmark) / the rest of that step decides / a later step does:
// counters[S3_CE_HIST_BASE + (dir * 3 + budget) * 3 + where].
#define S3_CE_QUEUE 128u                   // survivors a wave can hold between two verifications (fewer than 64 before a filter round, which adds at most 64)
#define S3_CE_QWORDS 7u
template <bool COUNT>
__global__ void __launch_bounds__(256) ceKernel(DevIndex ix, SearchIn in, SearchOut out, const uint32_t *taskRange)
{
    __shared__ uint4 sRec[4][64];                              // per wave: the task records of the batch ...
    __shared__ uint32_t sStart[4][65], sRead[4][64];           // ... the number of their first candidate; the read a task belongs to
    __shared__ uint4 sAux[4][64];                              // ... {its range's start in the reverse index, the read's 48 characters next to the matched part} (SearchOut::recAux)
    // the survivors' queue: {record's item word, first sequence number | flags, first suffix of the forward range, state << 8 | (case, strand) | size - 1 << 28 ..., read, k, sa}
    __shared__ uint32_t sQ[4][S3_CE_QWORDS][S3_CE_QUEUE];
    // (Until round 6 the tasks' reads were staged in LDS, one line per task: the filter's read characters now come with the task record, and the verification's
    //  survivors -- one candidate in ten -- read theirs from memory.  What the kernel is bound by, next to the lines it fetches, is the number of lines its
    //  wave-instructions TOUCH -- the vector cache looks up one line per lane and clock --: the text comes in aligned 16-byte pieces.)
    const uint32_t firstRec = taskRange[0];                   // records [firstRec, nRec): what the stepping launches wrote since the last call
    uint32_t nRec = taskRange[1];
    if (nRec > out.recCap) nRec = out.recCap;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t groups = (uint32_t)in.nCases * 2u;
    const uint32_t S = in.wpqLoad;
    const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t nSa = 0, nCe = 0, nAcc = 0;
    uint32_t histW[S3_CE_HIST_SLOTS] = {};                    // (COUNT) wave-uniform
    uint32_t recCur = 0, recEnd = 0;                          // wave-uniform: reserved block [recCur, recEnd)
    uint32_t qn = 0;                                          // wave-uniform: survivors in the queue

    auto histAdd = [&](uint32_t histIdx) {
        if constexpr (COUNT) {
#pragma unroll
            for (uint32_t k = 0; k < S3_CE_HIST_SLOTS; k++) histW[k] += (uint32_t)__popcll(__ballot(histIdx == k));
        }
    };
    auto histOf = [&](const S3Step &st0, int ins0, uint32_t where) -> uint32_t {
        const int left0 = st0.maxErr - ins0;
        return ((st0.dir > 0 ? 1u : 0u) * 3u + (uint32_t)(left0 < 0 ? 0 : left0 > 2 ? 2 : left0)) * 3u + where;
    };

    // ---- VERIFY: the first `take` survivors of the queue, one per lane ----
    auto verify = [&](uint32_t take) {
        bool hit = false;
        uint32_t rp = 0, recX = 0, seqK = 0, histIdx = 0xFFFFFFFFu;
        int acc = 0;
        if (lane < take) {
            recX = sQ[wv][0][lane];
            const uint32_t y = sQ[wv][1][lane], fl = sQ[wv][2][lane], w = sQ[wv][3][lane], rIdx = sQ[wv][4][lane], k = sQ[wv][5][lane];
            uint32_t sa = sQ[wv][6][lane];
            const uint32_t meta = (w >> 8) & 0xFFFFFu, sub = w & 0x7Fu, size = (w >> 28) + 1u + ((y >> 30) << 4);     // size - 1: four bits in w, two in y (<= 63: larger tasks never use a table)
            const bool haveSa = (w & 0x80u) != 0u;                                    // the flank record gave the position
            const int s0 = (int)(meta & 7u), i0 = (int)((meta >> 3) & 0x7FFu), ins0 = (int)((meta >> 14) & 7u), occ0 = (int)((meta >> 17) & 7u);
            const uint32_t casIdx = sub >> 1, strand = sub & 1u;
            const bool early = (y & S3_REC_EARLY) != 0;
            const int Li = in.uniformLen ? (int)in.uniformLen : (int)in.readLengths[rIdx];
            const int nSteps = (int)((in.nStepsPacked >> (3u * casIdx)) & 7u);
            const uint32_t *rd = in.readsT + (size_t)rIdx * 2u * S + (size_t)strand * S;
            const S3Case &cs = in.models[Li].cs[casIdx];
            const S3Step st0 = cs.st[s0];
            const int leftAlign = st0.dir > 0 ? st0.lm0 : st0.lm0 - i0;
            acc = occ0;
            uint32_t kf = k;                                                          // the candidate's place in the FORWARD range
            hit = true;
            if (!haveSa) sa = ix.sa[fl + k];
            else if (st0.dir > 0 && size > 1u) {
                // a forward task's candidates come in the reverse index's order: the survivor's place among the task's suffix-array entries (all requested at once)
                kf = 0xFFFFFFFFu;
                for (uint32_t j0 = 0; j0 < size; j0 += 8u) {
                    uint32_t v[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8u; j++) v[j] = j0 + j < size ? ix.sa[fl + j0 + j] : 0u;
#pragma unroll
                    for (uint32_t j = 0; j < 8u; j++) if (j0 + j < size && v[j] == sa) kf = j0 + j;
                }
                if (kf == 0xFFFFFFFFu) { atomicOr(out.flags, S3_FLAG_FLANK); hit = false; }
            }
            seqK = (y & 0xFFFFFFu) + kf;
            rp = sa - (uint32_t)leftAlign;                                            // 32-bit wrap, as the reference
            int t = s0, ii = i0, insCur = ins0;
            bool rejBound = false;
            while (hit && t < nSteps) {
                const S3Step st = cs.st[t];
                const int pos = st.start + st.dir * ii;
                const uint32_t left = (uint32_t)(pos < st.end ? pos : st.end), right = (uint32_t)(pos < st.end ? st.end : pos);
                const uint32_t len = right - left + 1u;
                if (early && t == s0) {
                    // what is left of the exact lookup step: the walk this replaces matches real text only
                    const long long rpS = (long long)sa - (long long)leftAlign;
                    if (rpS + (long long)left < 0 || rpS + (long long)right >= (long long)ix.n) { hit = false; rejBound = true; break; }
                }
                if (rp + left > ix.n) { hit = false; rejBound = true; break; }        // BGS-HostAlgnmtAlgo2.cpp:122-125
                const uint32_t tpos = rp + left;
                int mm = 0;
                // 64 characters at a time: all text and read words of the piece are requested before the first is used
                for (uint32_t o0 = 0; o0 < len; o0 += 64u) {
                    const uint32_t piece = len - o0 > 64u ? 64u : len - o0;
                    const uint32_t tp0 = tpos + o0, tw0 = tp0 >> 4, tsh = (tp0 & 15u) * 2u;
                    const uint32_t qp0 = left + o0, qw0 = qp0 >> 4, qsh = (qp0 & 15u) * 2u;
                    uint32_t tw[5], qw[5];
                    {
                        // the text words tw0 .. tw0 + 4 out of the aligned 16-byte pieces that hold them (the text ends in a zero tail longer than a piece:
                        // s3dp_api.hip), the first one at word `off` of the first piece
                        const uint32_t jmax = (piece + 15u) >> 4, off = tw0 & 3u, lastW = off + jmax;     // lastW <= 7
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 *tc = (const u32x4 *)ix.pac + (tw0 >> 2);
                        u32x4 c0 = tc[0], c1 = {0u, 0u, 0u, 0u};
                        if (lastW >= 4u) c1 = tc[1];
                        asm volatile("" : "+v"(c0), "+v"(c1));                    // (whole pieces: left alone the compiler asks for them word by word again)
                        const uint32_t W[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                        for (uint32_t j = 0; j < 5; j++) tw[j] = off == 0u ? W[j] : off == 1u ? W[j + 1] : off == 2u ? W[j + 2] : W[j + 3];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 5; j++) {
                        const uint32_t need = j * 16u < piece + 16u;                  // word j + the one behind it for the shift
                        if (!need) tw[j] = 0u;
                        qw[j] = (need && qw0 + j < S) ? rd[qw0 + j] : 0u;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        if (j * 16u < piece) {
                            const uint32_t cch = piece - j * 16u > 16u ? 16u : piece - j * 16u;
                            const uint32_t a_ = tsh ? (tw[j] << tsh) | (tw[j + 1] >> (32u - tsh)) : tw[j];
                            const uint32_t b_ = qsh ? (qw[j] << qsh) | (qw[j + 1] >> (32u - qsh)) : qw[j];
                            uint32_t x = a_ ^ b_;
                            x = (x | (x >> 1)) & 0x55555555u;
                            if (cch < 16u) x &= ~0u << (32u - 2u * cch);
                            mm += __popc(x);
                        }
                    }
                }
                nCe += (len + 31u) >> 5;
                const int tot = mm + insCur;
                if (tot < st.minErr || tot > st.maxErr) { hit = false; break; }
                acc += tot; t++; ii = 0; insCur = 0;
            }
            if constexpr (COUNT) {
                if (!hit) {
                    uint32_t where = 2u;                                         // a later step
                    if (t == s0) {
                        where = 1u;                                              // the rest of the first compared step (or the text's end)
                        const int pos = st0.start + st0.dir * i0;
                        const uint32_t left = (uint32_t)(pos < st0.end ? pos : st0.end), right = (uint32_t)(pos < st0.end ? st0.end : pos);
                        const uint32_t len = right - left + 1u, nAdj = len < 32u ? len : 32u;
                        // (a candidate no flank table served: would the 32 characters next to the matched part alone have decided?  VERDICT r5, item 5)
                        if (!haveSa && !rejBound && (st0.dir > 0 ? rp + left + nAdj <= ix.n : rp + right < ix.n)) {
                            int m2 = 0;
                            for (uint32_t q = 0; q < nAdj; q++) {
                                const uint32_t rpos = st0.dir > 0 ? left + q : right - q, tp = rp + rpos;
                                const uint32_t tc = (ix.pac[tp >> 4] >> (30u - 2u * (tp & 15u))) & 3u, rc = (rd[rpos >> 4] >> (30u - 2u * (rpos & 15u))) & 3u;
                                m2 += tc != rc ? 1 : 0;
                            }
                            if (m2 + ins0 > st0.maxErr) where = 0u;
                        }
                    }
                    histIdx = histOf(st0, ins0, where);
                }
            }
        }
        histAdd(histIdx);
        const unsigned long long hm = __ballot(hit);
        if (hm) {
            // accepted candidates go into a block of record slots the wave reserved with one atomic (as the stepping kernel does)
            const uint32_t np = (uint32_t)__popcll(hm);
            if (recCur + np > recEnd) {
                for (uint32_t j = recCur + lane; j < recEnd; j += 64u) out.recs[j] = make_uint4(0xFFFFFFFFu, 0, 0, 0);   // unused tail
                uint32_t b0 = 0;
                if (lane == 0) b0 = atomicAdd(out.recCount, S3_V2_REC_BLOCK);
                b0 = (uint32_t)__shfl((int)b0, 0);
                recCur = b0; recEnd = b0 + S3_V2_REC_BLOCK;
                if (recEnd > out.recCap || recEnd < b0) { recEnd = b0 < out.recCap ? out.recCap : b0; if (lane == 0) atomicOr(out.flags, S3_FLAG_REC_OVERFLOW); }
            }
            if (hit) {
                nAcc++;
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if (recCur + rk < recEnd) out.recs[recCur + rk] = make_uint4(recX, seqK | (1u << 24) | ((uint32_t)acc << 25), rp, rp);
            }
            recCur += np;
            if (recCur > recEnd) recCur = recEnd;
        }
        // what is left of the queue moves to its front
        __builtin_amdgcn_wave_barrier();
        uint32_t mv[S3_CE_QWORDS];
        const bool moves = take + lane < qn;
#pragma unroll
        for (uint32_t f = 0; f < S3_CE_QWORDS; f++) mv[f] = moves ? sQ[wv][f][take + lane] : 0u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t f = 0; f < S3_CE_QWORDS; f++) if (moves) sQ[wv][f][lane] = mv[f];
        qn -= take;
        __builtin_amdgcn_wave_barrier();
    };

    for (uint64_t base = (uint64_t)firstRec + (uint64_t)waveId * 64u; base < nRec; base += (uint64_t)nWaves * 64u) {
        const uint64_t idx = base + lane;
        uint4 rec = make_uint4(0xFFFFFFFFu, 0, 0, 0);
        if (idx < nRec) rec = out.recs[idx];
        bool task = rec.x != 0xFFFFFFFFu && (rec.y & S3_REC_TASK);
        uint32_t item = rec.x & ~S3_REC_PASS2;
        uint32_t slotOf = item / groups;
        if (task && (rec.x & S3_REC_PASS2)) {
            // second-pass records count in unit space: unit = heavy read * groups * U + item * U + position
            const uint32_t U = in.unitsPerItem;
            if (out.dynCap && item >= out.dynBase) item = out.dynOwner[item - out.dynBase];      // counted by an adopted frame: the unit its chain started from
            const uint32_t unit = item - in.unitBase;
            const uint32_t listed = unit / (groups * U);
            slotOf = out.heavyList[listed];
            item = slotOf * groups + (unit / U) % groups;
            // a task of a unit behind its read's cut: the units in front of it hold MaxOutputPerRead occurrences for sure, the merge never gets here
            if (out.readCut && unit - listed * groups * U > out.readCut[listed]) task = false;
        } else if (task && out.heavyFlags[slotOf]) task = false;                 // stale: the second pass searched this read again
        const uint32_t size = task ? (rec.w & 0xFFu) + 1u : 0u;
        // exclusive prefix sum of the sizes over the wave
        uint32_t incl = size;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, (int)d);
            if (lane >= d) incl += v;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        if (total == 0) continue;
        __builtin_amdgcn_wave_barrier();
        sStart[wv][lane] = incl - size;
        if (lane == 63u) sStart[wv][64] = total;
        const uint32_t myRead = task ? (in.active ? in.active[slotOf] : slotOf) : 0u;
        sRead[wv][lane] = myRead;
        sAux[wv][lane] = task ? out.recAux[idx] : make_uint4(0u, 0u, 0u, 0u);
        // A task of an item with a PADDED lookup key (S3_REC_PADDED, lookupPadded in s3dp_search_common.h) verifies the FORWARD range's suffixes, as the CPU
        // path does: its forward and reverse ranges need not hold the same occurrences (bit 7 of the task's low byte: no table for a forward step's candidates).
        // So does a task of more than 64 candidates (the queue's size field; the models' thresholds stop at 32).
        const uint32_t noTableBit = (size > 64u || ((rec.y & S3_REC_PADDED) && (rec.y & S3_REC_FWD))) ? 0x80u : 0u;
        sRec[wv][lane] = make_uint4(rec.x, rec.y, rec.z, (rec.w & 0xFFFFFF00u) | (item % groups) | noTableBit);     // (low byte: the (case, strand) of the item; bit 7: see above)
        __builtin_amdgcn_wave_barrier();
        // ---- FILTER: 64 candidates per round; a verification whenever the queue holds a wave's worth ----
        for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
            if (qn >= 64u) verify(64u);
            const uint32_t c = c0 + lane;
            bool survives = false;
            uint32_t q[S3_CE_QWORDS] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
            uint32_t histIdx = 0xFFFFFFFFu;
            if (c < total) {
                // the task this candidate belongs to: the last one that starts at or before c (tasks of size 0 start where the next one starts)
                uint32_t lo = 0, hi = 63u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (sStart[wv][mid] <= c) lo = mid; else hi = mid - 1u;
                }
                const uint4 tr = sRec[wv][lo];
                const uint32_t k = c - sStart[wv][lo], size = sStart[wv][lo + 1u] - sStart[wv][lo];
                const uint32_t meta = tr.w >> 8, sub = tr.w & 0x7Fu;
                const int s0 = (int)(meta & 7u), i0 = (int)((meta >> 3) & 0x7FFu), ins0 = (int)((meta >> 14) & 7u);
                const uint32_t casIdx = sub >> 1, rIdx = sRead[wv][lo];
                const int Li = in.uniformLen ? (int)in.uniformLen : (int)in.readLengths[rIdx];
                const bool fwdStep = (tr.y & S3_REC_FWD) != 0u;                  // (= the direction of step s0: said by the record, so that the flank record's load does not wait for the model's)
                const uint4 *tab = (tr.w & 0x80u) ? nullptr : fwdStep ? ix.flankFwd : ix.flankBack;
                const S3Step st0 = in.models[Li].cs[casIdx].st[s0];
                nSa++;
                survives = true;
                uint32_t sa = 0u;
                if (tab) {
                    const uint4 ax = sAux[wv][lo];
                    const uint4 e = tab[(fwdStep ? ax.x : tr.z) + k];
                    const int pos0 = st0.start + st0.dir * i0;
                    const uint32_t len0 = (uint32_t)(pos0 < st0.end ? st0.end - pos0 : pos0 - st0.end) + 1u, nAdj = len0 < 48u ? len0 : 48u;
                    const uint32_t q0 = ax.y, q1 = ax.z, q2 = ax.w;                    // the read's characters next to the matched part, the adjacent one first
                    auto miss = [](uint32_t a, uint32_t b, uint32_t n) -> int {        // mismatches among the first n <= 16 characters
                        uint32_t x = a ^ b;
                        x = (x | (x >> 1)) & 0x55555555u;
                        if (n < 16u) x &= n ? ~0u << (32u - 2u * n) : 0u;
                        return __popc(x);
                    };
                    const int mmAdj = miss(e.y, q0, nAdj < 16u ? nAdj : 16u) + miss(e.z, q1, nAdj <= 16u ? 0u : nAdj < 32u ? nAdj - 16u : 16u) +
                                      miss(e.w, q2, nAdj <= 32u ? 0u : nAdj - 32u);
                    // (only characters inside the text decide here: a window that leaves it is the full comparison's business, with its bounds checks)
                    const bool inText = fwdStep ? (uint64_t)e.x + nAdj <= (uint64_t)ix.n : e.x >= nAdj;
                    survives = !(inText && mmAdj + ins0 > st0.maxErr);
                    sa = fwdStep ? e.x - (uint32_t)(pos0 - st0.lm0) : e.x;
                    if constexpr (COUNT) { if (!survives) histIdx = histOf(st0, ins0, 0u); }
                }
                // (size - 1 <= 63: its low four bits ride in the state word's top, the other two in the sequence word's)
                q[0] = tr.x; q[1] = (tr.y & 0x3FFFFFFFu) | ((size - 1u) >> 4) << 30; q[2] = tr.z;
                q[3] = (tr.w & 0x0FFFFF00u) | sub | (tab ? 0x80u : 0u) | ((size - 1u) & 15u) << 28;
                q[4] = rIdx; q[5] = k; q[6] = sa;
            }
            histAdd(histIdx);
            const unsigned long long sm = __ballot(survives);
            if (survives) {
                const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
#pragma unroll
                for (uint32_t f = 0; f < S3_CE_QWORDS; f++) sQ[wv][f][at] = q[f];
            }
            qn += (uint32_t)__popcll(sm);
            __builtin_amdgcn_wave_barrier();
        }
    }
    while (qn) verify(qn < 64u ? qn : 64u);
    for (uint32_t j = recCur + lane; j < recEnd; j += 64u) out.recs[j] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (out.counters) {
        atomicAdd(&out.counters[2], (unsigned long long)nSa);
        atomicAdd(&out.counters[3], (unsigned long long)nCe);
        atomicAdd(&out.counters[10], (unsigned long long)nAcc);
        if constexpr (COUNT) {
            if (lane == 0) for (uint32_t k = 0; k < S3_CE_HIST_SLOTS; k++) if (histW[k]) atomicAdd(&out.counters[S3_CE_HIST_BASE + k], (unsigned long long)histW[k]);
        }
    }
}

void launchCheckExtend(const DevIndex &ix, const SearchIn &in, const SearchOut &out, const uint32_t *taskRange, hipStream_t s)
{
    if (in.nActive == 0) return;
    // the waves split the task records statically: launch exactly the blocks that are resident at once (a partial second round would run on a
    // third of the chip while the rest idles).  S3DP_CE_BLOCKS_PER_CU overrides the occupancy the runtime reports.
    const size_t lds = 0;
    static const unsigned cus = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return (unsigned)n;
    }();
    int perCu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, ceKernel<false>, 256, lds) != hipSuccess || perCu < 1) perCu = 4;
    if (const char *e = getenv("S3DP_CE_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) perCu = v; }
    const unsigned blocks = cus * (unsigned)perCu;
    if (out.counters) hipLaunchKernelGGL(ceKernel<true>, dim3(blocks), dim3(256), lds, s, ix, in, out, taskRange);
    else hipLaunchKernelGGL(ceKernel<false>, dim3(blocks), dim3(256), lds, s, ix, in, out, taskRange);
}

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadCe() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::ceKernel<false>); (void)hipGetLastError(); }
