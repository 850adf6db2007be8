// s3dp_search_v2.hip -- wave-cooperative 2BWT k-mismatch search for gfx950 (the production search kernel).
//
// Same results, record for record, as the reference CPU path; different machine mapping.  Measurement on MI355X (tools/hbm_probe.py) shows that a rank-query stream is bound
// by the number of vector-memory REQUESTS, not by DRAM bytes: two 16-B lane loads into the same 64-B line cost
// as much as two lines.  So v2 issues ONE coalesced 64-B request per rank line:
//
//   * the grid is persistent; a lane owns one READ at a time, walks its (case, strand) items back to back as a flat state machine and takes the
//     next read from a work queue when it is done.  Every trip of the loop first ISSUES all memory operations of all lanes (rank lines, lookup
//     entries, read words), then waits once, then every lane advances its own search (pure ALU + LDS) until it needs memory again;
//   * rank lines are fetched cooperatively: 4 adjacent lanes load the 4 x 16 B of one owner's line (the owner's
//     line id travels by ds_bpermute), the line is staged in LDS and read back by its owner -- "Occ/rank tables
//     staged through LDS for coalesced HBM reads".  l and r+1 almost always fall into the same line once the
//     range is narrow; only lanes whose two lines differ issue a private second fetch;
//   * the case's step table sits in LDS (one copy per block), the read's words -- both orientations, MSB first, prepared once per upload -- sit
//     in LDS word-interleaved by lane: a read character is one LDS word and a shift;
//   * an item starts from the longest lookup table the index holds (13-mers of the file format, or the 14- / 15- / 16-mer tables built in HBM at
//     upload: the state behind the lookup step's first characters, s3dp_kernels.hip::launchLongLookup);
//   * check-and-extend runs in its own kernel (ceKernel, below): the stepping kernel writes task records and goes on;
//   * a lane walks the (case, strand) items of its read one after the other -- fine for a read with a handful of candidates, a disaster for a read
//     inside a repeat family with 10^5 copies, whose 4-mismatch neighbourhood is 10^5 rank queries long: a few such lanes run for a second while
//     the chip idles.  So a read that reserves more candidates than in.heavyLimit leaves the first pass at once ("heavy" read, listed), and a
//     second launch of the same kernel takes the listed reads apart: its work unit is a slice of ONE (read, case, strand) item -- the part of the
//     item's search tree whose FIRST mismatch sits at one given position of the first step that allows mismatches (or that has none there).  The
//     depth-first order of the CPU path visits exactly these subtrees one after the other (at every position: the mismatch children first,
//     ascending, then the continuation), so the units of an item, numbered by position, ARE its sequence order: each unit counts its records
//     from zero and the counting scatter puts them behind each other -- a thousand lanes per heavy read, no sort.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "s3dp_device.h"
#include "s3dp_model.h"
#include "s3dp_search_common.h"

namespace s3 {

using namespace v2;

// Persistent wavefronts with per-lane refill: a lane owns one READ at a time and walks all its (case, strand) items
// back to back (the read is loaded once, as two 16-B requests from a per-read contiguous copy); when the read is
// finished the lane immediately takes the next one from the wave's chunk of the global work queue, so all 64 lanes
// keep a memory request in flight.
// UNIFORM: every read of the launch has the same length, the model (all cases) then lives in LDS; otherwise step
// descriptors are read from the global table (L1/L2 resident).
// Check-and-extend runs in its own kernel (ceKernel below).  A lane that reaches a small SA range inside the check window writes a
// TASK record (range + where it stands in the model) instead of walking the range itself, reserves one sequence number per candidate and goes
// on with its search: two dependent round trips per candidate (SA value, then text) and the compare code leave the stepping loop, which is
// bound by its instruction count.  The record stream keeps the CPU path's order because accepted candidates are written under the reserved
// numbers; rejected ones leave holes the ordering pass skips.  Not counting the candidates in `total` only delays the early stop at
// MaxOutputPerRead: the per-read cut is applied again, exactly, when the records are merged.
// SECOND (template): the second pass over the listed reads, in.unitMode = 1 / 2, two launches.  Its lanes do check-and-extend THEMSELVES (states
// ST_CE_SA / ST_CE_TEXT below) instead of writing task records: most occurrences of a read inside a repeat family are check-and-extend candidates,
// and only a lane that knows which of them hold can stop where the CPU path stops -- at MaxOutputPerRead occurrences.  The units of a read start in
// sequence order, spread over the launch (the work queue is ordered by item and position first, read last), every unit adds what it has found to
// counters the other units of its read can see, and a unit that finds MaxOutputPerRead occurrences in front of it stops.
//   1 (trunk pass): one lane per (listed read, item) walks the TRUNK of the item -- lookup step, exact steps, then the first mismatch step without
//     taking a mismatch -- and writes the range it holds at every position of that step into out.trunkStates; it then goes on as the item's last
//     unit (no mismatch in that step at all).  Whatever ends the trunk early (a range handed to check-and-extend, a dead range) ends it for all
//     units: positions it never reached keep an invalid state.
//   2 (unit pass): one lane per (listed read, item, position p): it loads the read and the trunk state of p, takes ONLY the mismatch children at p
//     and is from there on an ordinary depth-first search.
template <bool UNIFORM, bool SECOND>
__global__ void __launch_bounds__(S3_SEARCH_BLOCK, SECOND ? S3_V2_WAVES_PER_SIMD_SECOND : S3_V2_WAVES_PER_SIMD)
searchKernelV2(DevIndex ix, SearchIn in, SearchOut out)
{
    const uint32_t unitMode = SECOND ? in.unitMode : 0u;                    // (a constant 0 in the first pass: its code carries nothing of the second)
    extern __shared__ __align__(16) unsigned char shmem[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t wpq = in.wpqLoad;                                        // words per orientation in readsT (stride, a power of two)
    const uint32_t wl = in.wLds;                                            // words per orientation kept in LDS: what the longest searched length needs
    const uint32_t waveBytes = 4096u + 2u * wl * 256u;
    unsigned char *wbase = shmem + S3_V2_MODEL_BYTES + wv * waveBytes;
    uint4 *stage = (uint4 *)wbase;                                          // [64 owners][4 x 16 B]
    uint32_t *ro = (uint32_t *)(wbase + 4096u);                             // oriented read words, MSB first: ro[(strand * wl + w) * 64 + lane]
    uint32_t *myStage32 = (uint32_t *)(stage + lane * 4u);

    if (UNIFORM) {
        const uint32_t *src = (const uint32_t *)&in.models[in.uniformLen];
        for (uint32_t k = threadIdx.x; k < sizeof(S3Model) / 4; k += blockDim.x) ((uint32_t *)shmem)[k] = src[k];
        __syncthreads();
    }

    const uint32_t nCases = (uint32_t)in.nCases;
    const uint32_t groups = nCases * 2u;
    // second pass: listed reads x the items of this round (x the units of an item: every position of the mismatch step + "none")
    const uint32_t nListed = SECOND ? *in.slotCount : 0u;
    const uint32_t unitsPerReadHere = unitMode ? groups * (unitMode == 2u ? in.unitsPerItem : 1u) : 1u;
    const unsigned long long nSlots = unitMode ? (unsigned long long)nListed * unitsPerReadHere : in.nActive;
    uint32_t unitTrips = 0, maxUnitTrips = 0; unsigned long long laneTrips = 0;   // (statistics)
    uint32_t nCutSkip = 0, nCutMid = 0, nOwnMax = 0;                       // (statistics: units skipped by the cut before their first step / stopped by it / stopped by their own total)
    uint32_t nSaL = 0, nCeL = 0;                                           // second pass: suffix-array reads / compare windows of its own check-and-extend
    uint32_t nRank = 0, nLt = 0;

    // ---- per-lane item + search state ---------------------------------------------------------------------
    uint32_t state = ST_DONE;
    bool hasItem = false;
    uint32_t cntIdx = 0, seq = 0, total = 0, rIdx = 0, loadOff = 0, slotIdx = 0, sub = 0;
    uint32_t readTotal = 0;                                                 // occurrences the read's finished items have for sure (the cut at maxOut runs across items)
    uint32_t readCand = 0;                                                  // check-and-extend candidates this read has reserved
    uint32_t readRank = 0;                                                  // rank queries this read has cost so far (first pass)
    bool goHeavy = false;
    int Li = 0, nSteps = 0;
    uint32_t casIdx = 0;
    Range g = {0, 0, 0, 0};
    int s = 0, i = 0, ins = 0, occ = 0;
    // unit mode
    const uint32_t U = unitMode ? in.unitsPerItem : 1u;
    const uint32_t unitsPerItemHere = unitMode == 2u ? U : 1u;               // work units per item in this launch
    uint32_t listed = 0;                                                    // the lane's read in the list
    // check-and-extend of the second pass (one candidate at a time): candidates [ceL + ceK, ceL + ceN) of the suffix array are left; of the current
    // one: text position of the read's first character, the model step being compared, characters of it done, mismatches so far
    uint32_t ceL = 0, ceN = 0, ceK = 0, ceSa = 0, ceRp = 0;
    int ceT = 0, ceII = 0, ceIns = 0, ceAcc = 0, ceOff = 0, ceMm = 0;
    bool ceEarly = false;
    // A unit whose candidates keep failing stops verifying them itself: after S3_CE_GIVE_UP rejections in a row it writes task records like the first
    // pass (one trip per task instead of one or two per candidate; ceKernel verifies them, many lanes wide).  Knowing the count at once only pays
    // where candidates hold -- inside young repeat families, whose reads reach MaxOutputPerRead --, and either way the merge applies the cut exactly.
    uint32_t ceMiss = 0, ceTaken = 0;
    bool noCount = false;                                                   // unit mode: the lane found nothing to do (closed read, no trunk state): its counter is not its to write
    bool trunk = false;                                                     // the lane is still on the part of the item that all its units share
    uint32_t unitP = 0xFFFFFFFFu;                                           // position of the unit's first mismatch; 0xFFFFFFFF: none in the mismatch step
    int sStar = 0;                                                          // the first step of the case that allows a mismatch (nSteps: none)
    // A unit that holds MaxOutputPerRead certain occurrences by itself ends the read's list: every later unit of the read (in sequence order: item,
    // then position) lies behind the cut the merge will make.  It says so in out.readCut (smallest such unit per listed read); every lane of the second
    // pass reads that word along with its rank lines and stops when it finds itself behind it -- what the single lane of an item-wise walk gets from
    // its running total.
    uint32_t unitInRead = 0;
    const uint32_t *cutPtr = out.cutDummy;                                  // (a word of the batch nobody reads: a stray atomicMin must never land in the index)
    uint32_t strand = 0;
    Frame stack[STACK];
    int sp = 0;
    uint32_t ltA = 0, ltB = 0, ltC = 0, ltD = 0;                            // lookup-table keys of the starting item
    bool padded = false;                                                    // the item's lookup range may hold an element that is no occurrence
    // wave-uniform work queue window
    uint64_t chunkCur = 0, chunkEnd = 0, chunkBase = 0;
    uint32_t act0 = 0, act1 = 0, sl0 = 0, sl1 = 0;
    bool queueEmpty = false;

    auto stepAt = [&](int k) -> S3Step {
        if constexpr (UNIFORM) return ((const S3Model *)shmem)->cs[casIdx].st[k];      // LDS (address space known to the compiler)
        else return in.models[Li].cs[casIdx].st[k];                                    // global, L1/L2 resident
    };
    auto oword = [&](uint32_t w) -> uint32_t { return w < wl ? ro[(strand * wl + w) * 64u + lane] : 0u; };
    auto readChar = [&](int pos) -> uint32_t { return (ro[(strand * wl + ((uint32_t)pos >> 4)) * 64u + lane] >> (30u - 2u * ((uint32_t)pos & 15u))) & 3u; };
    auto readWindow = [&](int p0) -> uint32_t {               // 16 oriented characters from p0 >= 0, first one in the top bits
        const uint32_t w = (uint32_t)p0 >> 4, sh = ((uint32_t)p0 & 15u) * 2u;
        const uint32_t a = oword(w), b2 = oword(w + 1u);
        return sh ? (a << sh) | (b2 >> (32u - sh)) : a;
    };
    // A record is not written where it is found: the lane parks it (one per iteration) and the whole wave writes the
    // parked records at the end of the iteration into a block of record slots it reserved with ONE atomic per
    // S3_V2_REC_BLOCK records.  (Per-record atomics on one counter saturate the L2 atomic unit once lanes run
    // desynchronised: that was 20 M atomics per launch.)
    uint32_t pendMeta = 0, pendA = 0, pendB = 0;                // pendMeta bit 31 = valid
    uint32_t recCur = 0, recEnd = 0;                             // wave-uniform: reserved block [recCur, recEnd)
    auto emit = [&](uint32_t kind, uint32_t a, uint32_t b, uint32_t nmis) {
        pendMeta = 0x80000000u | seq | (kind << 24) | (nmis << 25);
        pendA = a; pendB = b;
        seq++;
    };
    // The running total of an item-wise walk, kept where the units of a read can all see it: per item (itemAcc) and, inside the item, per position
    // (the spare word of the trunk states: S3_TRUNK_VALID | what the unit at that position holds).  Everything in sequence order up to and including
    // this unit that adds up to MaxOutputPerRead puts the read's cut right behind this unit; the items up to this one adding up to it put it
    // behind the item.
    auto tsAt = [&](uint32_t q) -> uint4 * { return out.trunkStates + ((size_t)sub * U + q) * nListed + listed; };
    // (an accepted candidate is one occurrence: they are collected and booked S3_V2_COUNT_BATCH at a time, at the end of a range of candidates and at
    //  the end of the unit -- the evaluation reads up to groups + U words)
    uint32_t countPending = 0;
    auto countCertain = [&](uint32_t cnt) {
        cnt += countPending;
        countPending = 0;
        if (cnt == 0) return;
        uint32_t *acc = out.itemAcc + (size_t)listed * groups;
        atomicAdd(&acc[sub], cnt);
        uint32_t before = 0;
        for (uint32_t q = 0; q < sub; q++) before += __hip_atomic_load(&acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t mine = __hip_atomic_load(&acc[sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + mine >= in.maxOut) atomicMin((uint32_t *)cutPtr, sub * U + U - 1u);
        if (U > 1u) {
            const uint32_t myPos = unitInRead - sub * U;
            atomicAdd(&tsAt(myPos)->w, cnt);
            uint32_t upTo = before;
            for (uint32_t q = 0; q <= myPos; q++) upTo += __hip_atomic_load(&tsAt(q)->w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~S3_TRUNK_VALID;
            if (upTo >= in.maxOut) atomicMin((uint32_t *)cutPtr, unitInRead);
        } else if (total >= in.maxOut) atomicMin((uint32_t *)cutPtr, unitInRead);
    };
    // advance without memory until the lane needs ranks / SA / text or its item is finished
    auto settle = [&](bool needPop) {
        for (;;) {
            if (readTotal + total >= in.maxOut) { state = ST_DONE; return; }
            if (needPop) {
                if (sp == 0) { state = ST_DONE; return; }
                const Frame f = stack[--sp];
                g.fl = f.fl; g.fr = f.fr; g.rl = f.rl; g.rr = f.rl + (f.fr - f.fl);
                s = (int)(f.meta & 7u); i = (int)((f.meta >> 3) & 0x7FFu); ins = (int)((f.meta >> 14) & 7u); occ = (int)((f.meta >> 17) & 7u);
                needPop = false;
            }
            if (s >= nSteps) {                                               // SRA_STEP_TYPE_COMPLETE
                uint32_t l = g.fl, rr_ = g.fr;
                if (trunk && unitP != 0xFFFFFFFFu) { state = ST_DONE; return; }   // the whole trunk without a mismatch: the last unit's
                if (rr_ >= l) {
                    if (pendMeta >> 31) { state = ST_RESUME; return; }       // one record per lane and trip: this frame waits for the next trip
                    if (readTotal + total + (rr_ - l + 1u) > in.maxOut) rr_ = l + in.maxOut - readTotal - total - 1u;
                    emit(0, l, rr_, (uint32_t)(ins + occ));
                    total += rr_ - l + 1u;
                    if (unitMode) countCertain(rr_ - l + 1u);
                }
                needPop = true;                                              // the DFS goes on in the same trip
                continue;
            }
            const S3Step st = stepAt(s);
            const bool exact = ins >= st.maxErr;
            int lenEff = st.len;
            if (!exact && st.minErr > ins + 1) lenEff -= st.minErr - ins - 1;
            const bool alive = g.fl <= g.fr;
            if (i < lenEff && alive) {
                const int pos = st.start + st.dir * i;
                if ((g.fr - g.fl) < (uint32_t)st.ceThr && pos >= st.ceStart && pos <= st.ceEnd) {
                    if constexpr (SECOND) {
                        if (ceMiss < in.ceGiveUp) { ceL = g.fl; ceN = g.fr - g.fl + 1u; ceK = 0; ceEarly = false; state = ST_CE_SA; return; }   // the lane verifies them itself
                    }
                    if (pendMeta >> 31) { state = ST_RESUME; return; }       // one record per lane and trip
                    pendMeta = 0x80000000u | seq | S3_REC_TASK;
                    pendA = g.fl;
                    pendB = (g.fr - g.fl) | ((uint32_t)s | (uint32_t)i << 3 | (uint32_t)ins << 14 | (uint32_t)occ << 17) << 8;
                    seq += g.fr - g.fl + 1u;
                    readCand += g.fr - g.fl + 1u;
                    if (readCand > in.heavyLimit) { goHeavy = true; state = ST_DONE; return; }
                    needPop = true;
                    continue;
                }
                {
                    // Few suffixes left inside the exact lookup step: walking the rest of the step can only keep or lose each of them -- in the same
                    // relative order, exact steps map a range onto a range order-preservingly -- and with fewer than the next step's threshold the next step
                    // hands them to check-and-extend at its first position anyway (in.earlyCe: the largest range handed over; a candidate costs three random
                    // lines, a step one and it drops three candidates in four -- beyond ONE suffix walking is cheaper).  Hand them over now; the task also verifies what is left of this
                    // step, inside the text.  (Not for ranges that may hold a padded table element: those are walked, as the CPU path walks them.)
                    if (in.earlyCe && s == 0 && i >= (int)in.earlyMinI && st.ceThr == 0 && nSteps > 1 && !padded) {
                        const S3Step nx = stepAt(1);
                        const uint32_t lim = in.earlyCe < (uint32_t)nx.ceThr ? in.earlyCe : (uint32_t)nx.ceThr;     // nx.ceThr >= 0 for every step
                        const int nxLenEff = nx.minErr > 1 ? nx.len - (nx.minErr - 1) : nx.len;                    // the next step must have a first position to stand on
                        if ((g.fr - g.fl) < lim && nxLenEff > 0 && nx.start >= nx.ceStart && nx.start <= nx.ceEnd) {
                            if constexpr (SECOND) {
                                if (ceMiss < in.ceGiveUp) { ceL = g.fl; ceN = g.fr - g.fl + 1u; ceK = 0; ceEarly = true; state = ST_CE_SA; return; }
                            }
                            if (pendMeta >> 31) { state = ST_RESUME; return; }
                            pendMeta = 0x80000000u | seq | S3_REC_TASK | S3_REC_EARLY;
                            pendA = g.fl;
                            pendB = (g.fr - g.fl) | ((uint32_t)s | (uint32_t)i << 3 | (uint32_t)ins << 14 | (uint32_t)occ << 17) << 8;
                            seq += g.fr - g.fl + 1u;
                            readCand += g.fr - g.fl + 1u;
                            if (readCand > in.heavyLimit) { goHeavy = true; state = ST_DONE; return; }
                            needPop = true;
                            continue;
                        }
                    }
                }
                state = ST_STEP;
                return;
            }
            if (ins >= st.minErr && alive) {
                if (trunk && s == sStar) {
                    // past the last position of the mismatch step without a mismatch: the item's last unit ("none") goes on from here, in its turn
                    if (unitMode == 1u) *tsAt(U - 1u) = make_uint4(g.fl, g.fr, g.rl, S3_TRUNK_VALID);
                    state = ST_DONE;
                    return;
                }
                occ += ins; ins = 0; s++; i = 0;
                continue;
            }
            needPop = true;
        }
    };

    for (;;) {
        // ---------- retire finished items; next (case, strand) of the same read, or a new read ----------
        // (no global load in this section may stall the wave: step counts come packed in a kernel argument, the
        //  read index list of a whole chunk is prefetched into two registers per lane when the chunk is taken)
        if (state == ST_DONE && hasItem) {
            if (goHeavy) {
                // this read goes to the second pass: whatever it wrote so far is ignored there (scatter / check-and-extend test the flag)
                out.heavyFlags[slotIdx] = 1;
                {
                    const uint32_t at = atomicAdd(out.heavyCount, 1u);
                    out.heavyList[at] = slotIdx;
                    out.heavyIndex[slotIdx] = at;
                }
                pendMeta = 0;
                goHeavy = false;
                sub = groups;
            } else {
                // (the "none" unit's counter is shared by the trunk lane, which may have handed the trunk itself to check-and-extend, and the unit
                //  lane: a unit lane that wrote nothing -- closed, cut, no trunk state -- leaves the counter alone; it was zeroed before the pass)
                if constexpr (SECOND) countCertain(0u);
                if (!noCount && !(unitMode == 2u && seq == 0u)) out.itemCounts[cntIdx] = seq;
                readTotal += total;
                sub++;
                if (unitMode) sub = groups;                                 // second pass: one unit per turn
                else if (readTotal >= in.maxOut) {
                    // the per-read cut (mergeCountKernel) ends inside this item or before: what follows can never be kept
                    for (; sub < groups; sub++) out.itemCounts[slotIdx * groups + sub] = 0;
                }
            }
            if (sub < groups) {
                casIdx = sub >> 1; strand = sub & 1u;
                nSteps = (int)((in.nStepsPacked >> (3u * casIdx)) & 7u);
                sStar = (int)((in.sStarPacked >> (3u * casIdx)) & 7u);
                cntIdx = slotIdx * groups + sub;
                seq = 0; total = 0; sp = 0;
                state = ST_KEYS;
            } else hasItem = false;
        }
        const unsigned long long idleMask = __ballot(state == ST_DONE);
        if (idleMask) {
            if (chunkCur >= chunkEnd && !queueEmpty) {
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(out.workHead, (unsigned long long)S3_V2_CHUNK);
                b = (unsigned long long)__shfl((long long)b, 0);
                if (b >= nSlots) queueEmpty = true;
                else {
                    chunkCur = chunkBase = b; chunkEnd = b + S3_V2_CHUNK < nSlots ? b + S3_V2_CHUNK : nSlots;
                    if (in.slotList) {
                        sl0 = b + lane < nSlots ? in.slotList[(uint32_t)(b + lane) % nListed] : 0u;
                        sl1 = b + 64u + lane < nSlots ? in.slotList[(uint32_t)(b + 64u + lane) % nListed] : 0u;
                        if (in.active) { act0 = in.active[sl0]; act1 = in.active[sl1]; }
                    } else if (in.active) {
                        act0 = b + lane < nSlots ? in.active[b + lane] : 0u;
                        act1 = b + 64u + lane < nSlots ? in.active[b + 64u + lane] : 0u;
                    }
                }
            }
            const uint64_t avail = chunkEnd > chunkCur ? chunkEnd - chunkCur : 0;
            if (avail) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
                const uint32_t nIdle = (uint32_t)__popcll(idleMask);
                const uint32_t off = (uint32_t)(chunkCur - chunkBase) + rank;          // position inside the chunk, < 128 when taken
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((off & 63u) * 4u), (int)act0);
                const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((off & 63u) * 4u), (int)act1);
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((off & 63u) * 4u), (int)sl0);
                const uint32_t s1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((off & 63u) * 4u), (int)sl1);
                if (state == ST_DONE && rank < avail) {
                    slotIdx = (uint32_t)(chunkCur + rank);
                    sub = 0;
                    cntIdx = slotIdx * groups;
                    trunk = false; unitP = 0xFFFFFFFFu;
                    noCount = false;
                    ceMiss = 0; ceTaken = 0; unitTrips = 0;
                    if (unitMode) {
                        // queue order: item first, then position, the read last -- the units of one read start far apart, in their sequence order
                        const uint32_t rem = slotIdx / nListed;
                        listed = slotIdx - rem * nListed;
                        sub = rem / unitsPerItemHere;
                        const uint32_t p = unitMode == 2u ? rem % unitsPerItemHere : U - 1u;          // U - 1: the unit without a mismatch in the mismatch step
                        unitP = p + 1u == U ? 0xFFFFFFFFu : p;
                        cntIdx = in.unitBase + (listed * groups + sub) * U + p;
                        trunk = U > 1u && !(unitMode == 2u && p + 1u == U);
                        unitInRead = sub * U + p;
                        cutPtr = out.readCut + listed;
                        slotIdx = off < 64u ? s0 : s1;
                    }
                    rIdx = in.active ? (off < 64u ? a0 : a1) : slotIdx;
                    readTotal = 0; readCand = 0; readRank = 0;
                    if constexpr (UNIFORM) Li = (int)in.uniformLen;
                    casIdx = sub >> 1; strand = sub & 1u;
                    nSteps = (int)((in.nStepsPacked >> (3u * casIdx)) & 7u);
                    sStar = (int)((in.sStarPacked >> (3u * casIdx)) & 7u);
                    if (!unitMode) cntIdx += sub;
                    seq = 0; total = 0; sp = 0; loadOff = 0;
                    hasItem = true;
                    state = ST_LOAD;
                }
                chunkCur += nIdle < avail ? nIdle : avail;
            } else if (queueEmpty && idleMask == ~0ULL) break;                 // nothing left anywhere
        }

        // ---------- lookup-table keys of a starting item (BWTExactModel{Backward,Forward}_Lookup2, BGS-HostAlgnmtAlgo2.cpp:311-347) ----------
        if (state == ST_KEYS) {
            const S3Step st = stepAt(0);
            const int ll = ix.ltSize > st.len ? st.len : ix.ltSize;            // <= 13 characters = 26 bits of one window
            const uint32_t padOnes = ll < ix.ltSize ? ((1u << (2 * (ix.ltSize - ll))) - 1u) : 0u;
            const int shPad = 2 * (ix.ltSize - ll);
            {
                // does the item's 13-mer belong to the padded ones of the tables it reads?
                const int k0 = ix.ltSize;
                const uint32_t m13 = (1u << (2 * k0)) - 1u;
                padded = st.len < k0;                                        // shorter lookup steps read a whole block of entries: walked
                if (!padded) {
                    if (st.type == S3_T_BACK_LOOKUP) {
                        const uint32_t k13 = (readWindow(st.start - k0 + 1) >> (32 - 2 * k0)) & m13;
#pragma unroll
                        for (int j = 0; j < 12; j++) padded |= k13 == ix.padBack[j];
                    } else {
                        const uint32_t r13 = pairRev(readWindow(st.start)) & m13;
#pragma unroll
                        for (int j = 0; j < 24; j++) padded |= r13 == ix.padBi[j];
                    }
                }
            }
            if (ix.ltK && st.len >= ix.ltK && st.ceThr == 0) {                 // (a lookup step that may switch to check-and-extend on its way -- exact-match models -- is walked)
                // long table (<= 16 characters): one entry holds the state behind the lookup step's first ltK characters; ltB = ~0 marks the item
                const int K = ix.ltK;
                if (st.type == S3_T_BACK_LOOKUP) ltA = readWindow(st.start - K + 1) >> (32 - 2 * K);
                else ltA = pairRev(readWindow(st.start)) & (K >= 16 ? 0xFFFFFFFFu : (1u << (2 * K)) - 1u);
                ltB = 0xFFFFFFFFu;
            } else if (st.type == S3_T_BACK_LOOKUP) {
                const uint32_t key = readWindow(st.start - ll + 1) >> (32 - 2 * ll);
                ltA = key << shPad; ltB = (key << shPad) | padOnes;
            } else {
                const uint32_t key = readWindow(st.start) >> (32 - 2 * ll);
                ltA = key << shPad; ltB = (key << shPad) | padOnes;
                const uint32_t rkey = pairRev(readWindow(st.start)) & ((1u << (2 * ll)) - 1u);     // same characters, reversed order
                ltC = rkey << shPad; ltD = (rkey << shPad) | padOnes;
            }
            state = ST_LT;
        }
        // ---------- what does this lane need? (addresses only; no load may be consumed before the single wait) ----------
        uint32_t idA = ID_NONE, iL = 0, iR = 0;
        bool needB = false;
        int dir = 0;
        const uint4 *dummy16 = (const uint4 *)ix.pac;                // always-valid, cache-hot address for lanes that need nothing
        const uint4 *aP0 = dummy16, *aP1 = dummy16;
        const uint32_t *aD0 = ix.pac, *aD1 = ix.pac, *aD2 = ix.pac, *aD3 = ix.pac;
        if (state == ST_STEP) {
            dir = stepAt(s).dir;
            const uint32_t inv = dir < 0 ? ix.fwd.inverseSa0 : ix.rev.inverseSa0;
            iL = dir < 0 ? g.fl : g.rl;
            iR = (dir < 0 ? g.fr : g.rr) + 1u;
            iL -= (iL > inv);
            iR -= (iR > inv);
            const uint32_t kL = (iL + 63u) >> 7, kR = (iR + 63u) >> 7;
            idA = kL | (dir < 0 ? 0u : 0x80000000u);
            needB = kR != kL;
            if (unitMode) aD1 = cutPtr;
            if (needB) {
                const uint4 *line = (dir < 0 ? ix.fwd.lines : ix.rev.lines) + (size_t)kR * 4u;
                const bool fw = iR > (kR << 7);
                aP0 = line;
                aP1 = line + (fw ? 2 : 1);
                aD0 = (const uint32_t *)(line + 3) + (fw ? 1 : 0);
            }
        } else if (state == ST_LOAD) {
            const uint4 *src = (const uint4 *)(in.readsT + (size_t)rIdx * 2u * wpq + loadOff);   // wpq is a power of two
            aP0 = src;
            if (2u * wpq - loadOff > 4) aP1 = src + 1;
            if constexpr (!UNIFORM) aD0 = in.readLengths + rIdx;
        } else if (state == ST_TRUNK) {
            aP0 = tsAt(unitP == 0xFFFFFFFFu ? U - 1u : unitP);
            aD1 = cutPtr;
        } else if (SECOND && state == ST_CE_SA) {
            aD0 = ix.sa + ceL + ceK;
            aD1 = cutPtr;
        } else if (SECOND && state == ST_CE_TEXT) {
            // the 64-byte text line that holds the next character to compare; the next candidate's suffix-array entry rides along
            const S3Step st = stepAt(ceT);
            const int pos = st.start + st.dir * ceII;
            const uint32_t left = (uint32_t)(pos < st.end ? pos : st.end);
            idA = ID_TEXT | ((ceRp + left + (uint32_t)ceOff) >> 8);
            if (ceK + 1u < ceN) aD0 = ix.sa + ceL + ceK + 1u;
            aD1 = cutPtr;
        } else if (state == ST_LT) {
            const bool bidir = stepAt(0).type != S3_T_BACK_LOOKUP;
            if (ltB == 0xFFFFFFFFu) {
                if (bidir) aP0 = ix.ltBi + ltA;
                else { aD0 = (const uint32_t *)(ix.ltBack + ltA); aD1 = aD0 + 1; }
                nLt += 1;
            } else {
                aD0 = ix.lkt + (ltA ? ltA - 1u : 0u);
                aD1 = ix.lkt + ltB;
                if (bidir) { aD2 = ix.rlkt + (ltC ? ltC - 1u : 0u); aD3 = ix.rlkt + ltD; }
                nLt += bidir ? 4 : 2;
            }
        }
        // ---------- issue: every lane issues the same 10 loads, back to back, no branches ----------
        // cooperative rank lines: 4 adjacent lanes fetch the 4 x 16 B of one owner's line (owners without a request: line 0)
        const uint32_t sub = lane & 3u, q4 = lane >> 2;
        const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((q4) * 4u), (int)idA);
        const uint32_t o1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((16u + q4) * 4u), (int)idA);
        const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((32u + q4) * 4u), (int)idA);
        const uint32_t o3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((48u + q4) * 4u), (int)idA);
        auto lineAddr = [&](uint32_t o) -> const uint4 * {
            if (SECOND && o != ID_NONE && (o >> 30) == 1u) return (const uint4 *)ix.pac + (size_t)(o & 0x3FFFFFFFu) * 4u + sub;
            const uint4 *base = (o >> 31) ? ix.rev.lines : ix.fwd.lines;
            return base + (o == ID_NONE ? (size_t)sub : (size_t)(o & 0x7FFFFFFFu) * 4u + sub);
        };
        const uint4 va0 = *lineAddr(o0), va1 = *lineAddr(o1), va2 = *lineAddr(o2), va3 = *lineAddr(o3);
        const uint4 p0 = *aP0, p1 = *aP1;
        const uint32_t d0 = *aD0, d1 = *aD1, d2 = *aD2, d3 = *aD3;
        // ---------- stage the cooperative lines in LDS (owner slot = 64 B); slots of owners without a request get line 0 ----------
        stage[(q4) * 4u + sub] = va0;
        stage[(16u + q4) * 4u + sub] = va1;
        stage[(32u + q4) * 4u + sub] = va2;
        stage[(48u + q4) * 4u + sub] = va3;
        const uint32_t saVal = d0;
        __builtin_amdgcn_wave_barrier();

        // ---------- advance ----------
        bool doSettle = false, needPop = false;
        if constexpr (SECOND) { if (state != ST_DONE) { unitTrips++; laneTrips++; if (unitTrips > maxUnitTrips) maxUnitTrips = unitTrips; } }
        if (unitMode && (state == ST_STEP || state == ST_TRUNK || state == ST_CE_SA || state == ST_CE_TEXT) && unitInRead > d1) {    // behind the read's cut
            if (state == ST_TRUNK) nCutSkip++; else nCutMid++;
            state = ST_DONE; sp = 0;
        }
        // first pass: a read that has cost this many rank queries is heavy whatever it has found so far (dead ends cost steps, not candidates)
        if (!unitMode && state == ST_STEP && readRank > in.heavyRank) { goHeavy = true; state = ST_DONE; sp = 0; }
        if (state == ST_STEP) {
            const S3Step stS = stepAt(s);
            const bool exact = ins >= stS.maxErr;
            const RankSel qL = rankSel(iL), qR = rankSel(iR);
            const uint4 occA = stage[lane * 4u];
            // own line in LDS as words: [0..3] counts, [4..7] half before (reversed), [8..11] half after, [12] / [13] precounts
            const uint32_t offL = (qL.fwd ? 8u : 4u) + (qL.far ? 2u : 0u);
            const uint4 oL = rankFromWords(qL, occA, myStage32[offL], myStage32[offL + 1u], myStage32[12u + (qL.fwd ? 1u : 0u)]);
            uint4 oR;
            if (needB) oR = rankFromWords(qR, p0, qR.far ? p1.z : p1.x, qR.far ? p1.w : p1.y, d0);
            else {
                const uint32_t offR = (qR.fwd ? 8u : 4u) + (qR.far ? 2u : 0u);
                oR = rankFromWords(qR, occA, myStage32[offR], myStage32[offR + 1u], myStage32[12u + (qR.fwd ? 1u : 0u)]);
            }
            nRank += 2;
            readRank += 2;
            const int pos = stS.start + stS.dir * i;
            const uint32_t ch = readChar(pos);
            // child ranges: the active range advances by rank, the passive one shrinks by the occurrences of the larger symbols
            const uint32_t d1 = oR.y - oL.y, d2 = oR.z - oL.z, d3 = oR.w - oL.w;
            const uint32_t base = stS.dir < 0 ? g.rr : g.fr;
            auto child = [&](uint32_t al, uint32_t ar, uint32_t oc) -> Range {
                const uint32_t pr = base - oc, pl = pr - (ar - al);
                Range o;
                if (stS.dir < 0) { o.fl = al; o.fr = ar; o.rl = pl; o.rr = pr; }
                else { o.rl = al; o.rr = ar; o.fl = pl; o.fr = pr; }
                return o;
            };
            const Range gc = child((ch == 0 ? oL.x : ch == 1 ? oL.y : ch == 2 ? oL.z : oL.w) + 1u, ch == 0 ? oR.x : ch == 1 ? oR.y : ch == 2 ? oR.z : oR.w,
                                   ch == 0 ? d1 + d2 + d3 : ch == 1 ? d2 + d3 : ch == 2 ? d3 : 0u);
            doSettle = true;
            const bool onTrunkHere = trunk && s == sStar;                      // trunk pass: no mismatch is taken in this step ...
            const bool branchOnly = onTrunkHere && (uint32_t)i == unitP;        // ... unit pass, at the unit's position: only the mismatch children
            if (branchOnly) trunk = false;
            else if (onTrunkHere && (uint32_t)i + 1u < U) *tsAt((uint32_t)i) = make_uint4(g.fl, g.fr, g.rl, S3_TRUNK_VALID);
            if (!exact && !(onTrunkHere && !branchOnly)) {
                const bool a0 = ch != 0 && oL.x + 1u <= oR.x, a1 = ch != 1 && oL.y + 1u <= oR.y;
                const bool a2 = ch != 2 && oL.z + 1u <= oR.z, a3 = ch != 3 && oL.w + 1u <= oR.w;
                if (branchOnly && !(a0 || a1 || a2 || a3)) { needPop = true; }   // no child alive: the unit is empty (sp == 0: the pop ends it)
                else if (a0 || a1 || a2 || a3) {
                    if (sp + 4 > STACK) { atomicOr(out.flags, S3_FLAG_STACK_OVERFLOW); state = ST_DONE; sp = 0; doSettle = false; }
                    else {
                        const uint32_t mCont = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)ins << 14) | ((uint32_t)occ << 17);
                        const uint32_t mKid = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)(ins + 1) << 14) | ((uint32_t)occ << 17);
                        Frame f;
                        f.fl = gc.fl; f.fr = gc.fr; f.rl = gc.rl; f.meta = mCont;
                        if (!branchOnly) stack[sp++] = f;                    // (the continuation past the unit's position belongs to the later units)
                        f.meta = mKid;
                        if (a3) { const Range k = child(oL.w + 1u, oR.w, 0u); f.fl = k.fl; f.fr = k.fr; f.rl = k.rl; stack[sp++] = f; }
                        if (a2) { const Range k = child(oL.z + 1u, oR.z, d3); f.fl = k.fl; f.fr = k.fr; f.rl = k.rl; stack[sp++] = f; }
                        if (a1) { const Range k = child(oL.y + 1u, oR.y, d2 + d3); f.fl = k.fl; f.fr = k.fr; f.rl = k.rl; stack[sp++] = f; }
                        if (a0) { const Range k = child(oL.x + 1u, oR.x, d1 + d2 + d3); f.fl = k.fl; f.fr = k.fr; f.rl = k.rl; stack[sp++] = f; }
                        needPop = true;
                    }
                }
            }
            if (doSettle && !needPop) { g = gc; i++; }
        }
        if (state == ST_RESUME) { doSettle = true; needPop = false; }          // the completed frame that had to wait: emit it now, then pop
        if (state == ST_TRUNK) {
            // the range the trunk pass held at this unit's position (.w = 0: the trunk never got there -- nothing to do)
            if (!(p0.w & S3_TRUNK_VALID)) { state = ST_DONE; noCount = true; }
            else {
                g.fl = p0.x; g.fr = p0.y; g.rl = p0.z; g.rr = p0.z + (p0.y - p0.x); s = sStar; ins = 0; occ = 0;
                if (unitP != 0xFFFFFFFFu) { i = (int)unitP; state = ST_STEP; }
                else { i = (int)stepAt(sStar).len; doSettle = true; needPop = false; }      // the "none" unit: behind the step's last position
            }
        } else
        if (state == ST_LOAD) {
            // word gi of the read's row in readsT (both orientations, wpq words each) -> LDS word (orientation * wl + w); words past wl are padding
            const uint32_t vals[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t gi = loadOff + k;
                const uint32_t so = gi >= wpq ? 1u : 0u, w = gi - so * wpq;
                if (gi < 2u * wpq && w < wl) ro[(so * wl + w) * 64u + lane] = vals[k];
            }
            loadOff += 8;
            if constexpr (!UNIFORM) Li = (int)saVal;
            if (loadOff >= 2u * wpq) state = unitMode == 2u ? ST_TRUNK : ST_KEYS;
        } else if (state == ST_LT) {
            const S3Step st = stepAt(0);
            int ll = ix.ltSize > st.len ? st.len : ix.ltSize;
            if (ltB == 0xFFFFFFFFu) {
                ll = ix.ltK;
                if (st.type == S3_T_BACK_LOOKUP) { g.fl = d0; g.fr = d1; g.rl = g.fl; g.rr = g.fr; }
                else { g.fl = p0.x; g.fr = p0.y; g.rl = p0.z; g.rr = p0.w; }
            } else if (st.type == S3_T_BACK_LOOKUP) {
                g.fl = ltA ? d0 + 1u : 1u; g.fr = d1; g.rl = g.fl; g.rr = g.fr;
            } else {
                uint32_t l = ltA ? d0 + 1u : 1u, rr_ = d1, rl = ltC ? d2 + 1u : 1u, rrr = d3;
                if (rr_ - l < rrr - rl) rl = rrr - (rr_ - l);                          // :349-356
                else if (rr_ - l > rrr - rl) l = rr_ - (rrr - rl);
                g.fl = l; g.fr = rr_; g.rl = rl; g.rr = rrr;
            }
            s = 0; i = ll; ins = 0; occ = 0;
            doSettle = true;
        }
        if constexpr (SECOND) {
            // ---------- check-and-extend by the lane itself: per candidate its suffix-array entry, then the text under the rest of the model, one 64-byte
            // line (256 characters) per trip, compared as ceKernel compares (XOR, fold, popcount); BGS-HostAlgnmtAlgo2.cpp:86-169 ----------
            if (state == ST_CE_SA || state == ST_CE_TEXT) {
                bool textHere = state == ST_CE_TEXT;                          // the line of this trip belongs to the current candidate
                bool fresh = !textHere;                                       // the current candidate has not been set up yet
                const bool haveNext = textHere && ceK + 1u < ceN;             // ... and the next candidate's entry came along
                const uint32_t lineHere = idA & 0x3FFFFFFFu;
                if (fresh) ceSa = saVal;
                for (;;) {
                    int verdict = 0;                                          // 1: accepted, -1: rejected, 0: needs a text line
                    if (fresh) {
                        const S3Step st0 = stepAt(s);
                        ceRp = ceSa - (uint32_t)(st0.dir > 0 ? st0.lm0 : st0.lm0 - i);     // 32-bit wrap, as the reference
                        ceT = s; ceII = i; ceIns = ins; ceAcc = occ; ceOff = 0; ceMm = 0;
                        nSaL++;
                        fresh = false;
                    }
                    for (;;) {
                        const S3Step st = stepAt(ceT);
                        const int pos = st.start + st.dir * ceII;
                        const uint32_t left = (uint32_t)(pos < st.end ? pos : st.end), right = (uint32_t)(pos < st.end ? st.end : pos);
                        const uint32_t len = right - left + 1u;
                        if (ceOff == 0) {
                            // entering a step: its piece must lie inside the text
                            if (ceEarly && ceT == s) {
                                const long long rpS = (long long)ceSa - (long long)(st.dir > 0 ? st.lm0 : st.lm0 - i);
                                if (rpS + (long long)left < 0 || rpS + (long long)right >= (long long)ix.n) { verdict = -1; break; }
                            }
                            if (ceRp + left > ix.n) { verdict = -1; break; }    // BGS-HostAlgnmtAlgo2.cpp:122-125
                        }
                        const uint32_t tpos = ceRp + left + (uint32_t)ceOff;
                        if (!textHere || (tpos >> 8) != lineHere) break;        // that line is not here: next trip
                        uint32_t todo = len - (uint32_t)ceOff;
                        const uint32_t inLine = 256u - (tpos & 255u);
                        if (todo > inLine) todo = inLine;
                        for (uint32_t o = 0; o < todo; o += 16u) {
                            const uint32_t c = todo - o > 16u ? 16u : todo - o;
                            const uint32_t tp = (tpos + o) & 255u, w = tp >> 4, sh = (tp & 15u) * 2u;
                            const uint32_t w0 = myStage32[w], w1 = w + 1u < 16u ? myStage32[w + 1u] : 0u;
                            const uint32_t a_ = sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
                            uint32_t x = a_ ^ readWindow((int)(left + (uint32_t)ceOff + o));
                            x = (x | (x >> 1)) & 0x55555555u;
                            if (c < 16u) x &= ~0u << (32u - 2u * c);
                            ceMm += __popc(x);
                        }
                        nCeL += (todo + 31u) >> 5;
                        ceOff += (int)todo;
                        if ((uint32_t)ceOff < len) break;                          // the rest of the step lies in the next line
                        const int tot = ceMm + ceIns;
                        if (tot < st.minErr || tot > st.maxErr) { verdict = -1; break; }
                        ceAcc += tot; ceT++; ceII = 0; ceIns = 0; ceOff = 0; ceMm = 0;
                        if (ceT >= nSteps) { verdict = 1; break; }
                    }
                    if (verdict == 0) { state = ST_CE_TEXT; break; }
                    if (verdict > 0) {
                        // an occurrence: written under the unit's next sequence number (rejected candidates reserve nothing here)
                        pendMeta = 0x80000000u | seq | (1u << 24) | ((uint32_t)ceAcc << 25);
                        pendA = ceRp; pendB = ceRp;
                        seq++; total++;
                        if (++countPending >= S3_V2_COUNT_BATCH) countCertain(0u);
                        // (a unit that has verified in.ceInlineMax occurrences itself has shown what it is worth to its read's cut: from there on its
                        //  candidates go to ceKernel too -- a thousand of them one after the other is what a launch would wait for)
                        ceMiss = ++ceTaken >= in.ceInlineMax ? in.ceGiveUp : 0u;
                    } else ceMiss++;
                    ceK++;
                    if (ceK >= ceN) { countCertain(0u); doSettle = true; needPop = true; break; }  // all candidates seen: the search goes on (settle sets the state)
                    if (ceMiss >= in.ceGiveUp && !(pendMeta >> 31)) {
                        // the rest of this range goes to ceKernel as a task (its candidates reserve their sequence numbers, as in the first pass)
                        pendMeta = 0x80000000u | seq | S3_REC_TASK | (ceEarly ? S3_REC_EARLY : 0u);
                        pendA = ceL + ceK;
                        pendB = (ceN - ceK - 1u) | ((uint32_t)s | (uint32_t)i << 3 | (uint32_t)ins << 14 | (uint32_t)occ << 17) << 8;
                        seq += ceN - ceK;
                        doSettle = true; needPop = true;
                        break;
                    }
                    if (readTotal + total >= in.maxOut) { state = ST_DONE; sp = 0; nOwnMax++; break; }
                    if (textHere && haveNext) { ceSa = saVal; fresh = true; textHere = false; continue; }   // set the next one up now; its text comes with the next trip
                    state = ST_CE_SA;
                    break;
                }
            }
        }
        if (doSettle) settle(needPop);
        // ---------- write the parked records (convergent) ----------
        {
            const unsigned long long pm = __ballot((pendMeta >> 31) != 0);
            if (pm) {
                const uint32_t np = (uint32_t)__popcll(pm);
                if (recCur + np > recEnd) {
                    for (uint32_t k = recCur + lane; k < recEnd; k += 64u) out.recs[k] = make_uint4(0xFFFFFFFFu, 0, 0, 0);   // unused tail
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(out.recCount, S3_V2_REC_BLOCK);
                    b = (uint32_t)__shfl((int)b, 0);
                    recCur = b; recEnd = b + S3_V2_REC_BLOCK;
                    if (recEnd > out.recCap || recEnd < b) { recEnd = b < out.recCap ? out.recCap : b; atomicOr(out.flags, S3_FLAG_REC_OVERFLOW); }
                }
                if (pendMeta >> 31) {
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    if (recCur + rk < recEnd) out.recs[recCur + rk] = make_uint4(unitMode ? cntIdx | S3_REC_PASS2 : cntIdx, pendMeta & 0x7FFFFFFFu, pendA, pendB);
                    pendMeta = 0;
                }
                recCur += np;
                if (recCur > recEnd) recCur = recEnd;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t k = recCur + lane; k < recEnd; k += 64u) out.recs[k] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (out.counters) {
        atomicAdd(&out.counters[0], (unsigned long long)nRank);
        atomicAdd(&out.counters[1], (unsigned long long)nLt);
        if constexpr (SECOND) {
            atomicAdd(&out.counters[2], (unsigned long long)nSaL);
            atomicAdd(&out.counters[3], (unsigned long long)nCeL);
            atomicAdd(&out.counters[4], (unsigned long long)nCutSkip);
            atomicAdd(&out.counters[5], (unsigned long long)nCutMid);
            atomicAdd(&out.counters[6], laneTrips);
            atomicMax(&out.counters[7], (unsigned long long)maxUnitTrips);
        }
    }
}

// ---- check-and-extend of the task records --------------------------------------------------------------------------------------------------
// A wave takes 64 records at a time; the candidates of their tasks are numbered through (wave prefix sum of the task sizes, descriptors in LDS) and
// dealt to the lanes 64 at a time -- one lane per CANDIDATE, whatever the sizes of the tasks (1 .. 256): a lane finds its task by bisection in the
// prefix sums.  Accepted candidates are appended to the record stream with one atomic per wave and S3_V2_REC_BLOCK records.  Per candidate: SA
// value, then per remaining model step the text against the oriented read, 16 characters at a time (XOR, fold, popcount), the step's error
// window checked as in the stepping kernel (BGS-HostAlgnmtAlgo2.cpp:86-169).
__global__ void __launch_bounds__(256) ceKernel(DevIndex ix, SearchIn in, SearchOut out, const uint32_t *taskRange)
{
    __shared__ uint4 sRec[4][64];                              // per wave: the task records of the batch ...
    __shared__ uint32_t sStart[4][65], sRead[4][64];           // ... the number of their first candidate; the read a task belongs to
    const uint32_t firstRec = taskRange[0];                   // records [firstRec, nRec): what the stepping launches wrote since the last call
    uint32_t nRec = taskRange[1];
    if (nRec > out.recCap) nRec = out.recCap;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t groups = (uint32_t)in.nCases * 2u;
    const uint32_t S = in.wpqLoad;
    const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t nSa = 0, nCe = 0;
    uint32_t recCur = 0, recEnd = 0;                          // wave-uniform: reserved block [recCur, recEnd)
    for (uint64_t base = (uint64_t)firstRec + (uint64_t)waveId * 64u; base < nRec; base += (uint64_t)nWaves * 64u) {
        const uint64_t idx = base + lane;
        uint4 rec = make_uint4(0xFFFFFFFFu, 0, 0, 0);
        if (idx < nRec) rec = out.recs[idx];
        bool task = rec.x != 0xFFFFFFFFu && (rec.y & S3_REC_TASK);
        uint32_t item = rec.x & ~S3_REC_PASS2;
        uint32_t slotOf = item / groups;
        if (task && (rec.x & S3_REC_PASS2)) {
            // second-pass records (units that gave up verifying their candidates themselves) count in unit space:
            // unit = heavy read * groups * U + item * U + position
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
        sRec[wv][lane] = make_uint4(rec.x, rec.y, rec.z, (rec.w & 0xFFFFFF00u) | (item % groups));     // (low byte: the (case, strand) of the item)
        sStart[wv][lane] = incl - size;
        if (lane == 63u) sStart[wv][64] = total;
        sRead[wv][lane] = task ? (in.active ? in.active[slotOf] : slotOf) : 0u;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
            const uint32_t c = c0 + lane;
            bool hit = false;
            uint32_t rp = 0, recX = 0, seqK = 0;
            int acc = 0;
            if (c < total) {
                // the task this candidate belongs to: the last one that starts at or before c (tasks of size 0 start where the next one starts)
                uint32_t lo = 0, hi = 63u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (sStart[wv][mid] <= c) lo = mid; else hi = mid - 1u;
                }
                const uint4 tr = sRec[wv][lo];
                const uint32_t k = c - sStart[wv][lo];
                const uint32_t meta = tr.w >> 8, sub = tr.w & 0xFFu;
                const int s0 = (int)(meta & 7u), i0 = (int)((meta >> 3) & 0x7FFu), ins0 = (int)((meta >> 14) & 7u), occ0 = (int)((meta >> 17) & 7u);
                const uint32_t casIdx = sub >> 1, strand = sub & 1u, rIdx = sRead[wv][lo];
                const bool early = (tr.y & S3_REC_EARLY) != 0;
                const int Li = in.uniformLen ? (int)in.uniformLen : (int)in.readLengths[rIdx];
                const int nSteps = (int)((in.nStepsPacked >> (3u * casIdx)) & 7u);
                const uint32_t *rd = in.readsT + (size_t)rIdx * 2u * S + (size_t)strand * S;
                const S3Case &cs = in.models[Li].cs[casIdx];
                recX = tr.x; seqK = (tr.y & 0xFFFFFFu) + k;
                acc = occ0;
                const uint32_t sa = ix.sa[tr.z + k];
                nSa++;
                const S3Step st0 = cs.st[s0];
                const int leftAlign = st0.dir > 0 ? st0.lm0 : st0.lm0 - i0;
                rp = sa - (uint32_t)leftAlign;                                   // 32-bit wrap, as the reference
                int t = s0, ii = i0, insCur = ins0;
                hit = true;
                while (t < nSteps) {
                    const S3Step st = cs.st[t];
                    const int pos = st.start + st.dir * ii;
                    const uint32_t left = (uint32_t)(pos < st.end ? pos : st.end), right = (uint32_t)(pos < st.end ? st.end : pos);
                    const uint32_t len = right - left + 1u;
                    if (early && t == s0) {
                        // what is left of the exact lookup step: the walk this replaces matches real text only
                        const long long rpS = (long long)sa - (long long)leftAlign;
                        if (rpS + (long long)left < 0 || rpS + (long long)right >= (long long)ix.n) { hit = false; break; }
                    }
                    if (rp + left > ix.n) { hit = false; break; }                // BGS-HostAlgnmtAlgo2.cpp:122-125
                    const uint32_t tpos = rp + left;
                    int mm = 0;
                    // 128 characters at a time: all text and read words of the piece are requested before the first is used
                    for (uint32_t o0 = 0; o0 < len; o0 += 128u) {
                        const uint32_t piece = len - o0 > 128u ? 128u : len - o0;
                        const uint32_t tp0 = tpos + o0, tw0 = tp0 >> 4, tsh = (tp0 & 15u) * 2u;
                        const uint32_t qp0 = left + o0, qw0 = qp0 >> 4, qsh = (qp0 & 15u) * 2u;
                        uint32_t tw[9], qw[9];
#pragma unroll
                        for (uint32_t j = 0; j < 9; j++) {
                            const uint32_t need = j * 16u < piece + 16u;              // word j + the one behind it for the shift
                            tw[j] = (need && tw0 + j < ix.pacWords) ? ix.pac[tw0 + j] : 0u;
                            qw[j] = (need && qw0 + j < S) ? rd[qw0 + j] : 0u;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
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
            }
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
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                    if (recCur + rk < recEnd) out.recs[recCur + rk] = make_uint4(recX, seqK | (1u << 24) | ((uint32_t)acc << 25), rp, rp);
                }
                recCur += np;
                if (recCur > recEnd) recCur = recEnd;
            }
        }
    }
    for (uint32_t j = recCur + lane; j < recEnd; j += 64u) out.recs[j] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (out.counters) {
        atomicAdd(&out.counters[2], (unsigned long long)nSa);
        atomicAdd(&out.counters[3], (unsigned long long)nCe);
    }
}

// First pass (in.slotList == NULL): one work unit per read of in.nActive slots.  Second pass: one unit per (listed read, item); the kernel reads
// the number of listed reads from device memory, so the launch needs no host round trip.  Returns a hipError_t.
int launchSearchV2(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s)
{
    if (in.nActive == 0) return (int)hipSuccess;
    uint64_t blocks = ((uint64_t)in.nActive * (in.unitMode ? (uint64_t)in.nCases * 2u * (in.unitMode == 2u ? in.unitsPerItem : 1u) : 1u) + S3_SEARCH_BLOCK - 1) / S3_SEARCH_BLOCK;
    if (blocks > maxBlocks) blocks = maxBlocks;
    const size_t lds = S3_V2_MODEL_BYTES + (size_t)(S3_SEARCH_BLOCK / 64) * (4096u + 2u * in.wLds * 256u);
    if (lds > 48u * 1024u) {
        // more than the default dynamic LDS limit: without the attribute the launch itself fails later with a less useful error
        hipError_t e = hipFuncSetAttribute((const void *)searchKernelV2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)searchKernelV2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)searchKernelV2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)searchKernelV2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((unsigned)blocks), block(S3_SEARCH_BLOCK);
    if (in.unitMode) {
        if (in.uniformLen) hipLaunchKernelGGL((searchKernelV2<true, true>), grid, block, lds, s, ix, in, out);
        else hipLaunchKernelGGL((searchKernelV2<false, true>), grid, block, lds, s, ix, in, out);
    } else if (in.uniformLen) hipLaunchKernelGGL((searchKernelV2<true, false>), grid, block, lds, s, ix, in, out);
    else hipLaunchKernelGGL((searchKernelV2<false, false>), grid, block, lds, s, ix, in, out);
    return (int)hipSuccess;
}

void launchCheckExtend(const DevIndex &ix, const SearchIn &in, const SearchOut &out, const uint32_t *taskRange, hipStream_t s)
{
    if (in.nActive == 0) return;
    // the waves split the task records statically: launch exactly the blocks that are resident at once (a partial second round would run on a
    // third of the chip while the rest idles).  S3DP_CE_BLOCKS_PER_CU overrides the occupancy the runtime reports.
    static const unsigned blocks = [] {
        int dev = 0, cus = 256, perCu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, ceKernel, 256, 0) != hipSuccess || perCu < 1) perCu = 6;
        if (const char *e = getenv("S3DP_CE_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) perCu = v; }
        return (unsigned)(cus * perCu);
    }();
    hipLaunchKernelGGL(ceKernel, dim3(blocks), dim3(256), 0, s, ix, in, out, taskRange);
}

}  // namespace s3
