// s3dp_unit.hip -- the UNIT PASS of the 2BWT k-mismatch search for gfx950: the listed ("heavy") reads of a search pass, taken apart.
//
// What a unit is, and why the units of an item in position order ARE the CPU path's record order, is told in s3dp_search_common.h (s3dp_first.hip runs the
// first pass and the trunk pass).  This file is the kernel that searches the units: one lane per (listed read, item, GROUP of positions of the item's first
// mismatch: two for single-end ladders, one for pairs -- "GROUPS" below).  Until round 3 it was an instantiation of the one search kernel and carried everything that kernel could do -- read loading for both strands, lookup
// tables, key building, trunk bookkeeping, ten unconditional loads per lane and trip, the depth-first stack in scratch memory whose pops stalled
// the wave in the middle of a trip -- at 153 registers and three waves per SIMD, and a launch lasted as long as its longest unit.  Here:
//
//   * a wave takes 64 queue slots at a time and TESTS them before any lane starts one: the slots of a chunk are consecutive listed reads at one
//     (item, position), so their trunk states, cut words and read numbers are three coalesced loads; units without a trunk state or behind their
//     read's cut (half of them) never occupy a lane.  The queue is worked through a two-stage pipeline inside the trips' single wait: the atomic
//     on the queue head one trip, the loads of the chunk the next;
//   * a unit needs ONE strand of its read in LDS (a single trip);
//   * the step descriptor of the model step a lane is in sits in four registers;
//   * the depth-first stack is a ring of frames in global memory, [frame][lane] so that the lanes of a wave write whole lines, with the TOP frame
//     always in registers: a push is a store nobody waits for, a pop takes the register copy, and the frame below it is fetched with the next
//     trip's rank lines;
//   * the read's cut word is read when a lane pops (and while it verifies candidates), not on every trip;
//   * occurrences are booked S3_V2_COUNT_BATCH at a time (the evaluation of the cut reads up to groups + U words);
//   * loads are issued only by the lanes whose state needs them;
//   * and the tail: once the queue is empty, an idle lane ADOPTS the oldest frame of a busy lane of its wave -- the subtree that lane would have
//     searched last.  Its records count from zero in a counter of their own that is linked right behind the giver's counter (SearchOut::chainNext);
//     unitChainKernel turns the chains into offsets and the counting scatter places the records exactly where the single lane would have put
//     them.  A unit of ten thousand trips becomes a few hundred trips of a whole wave; the order of the records does not change;
//   * and since round 5: FAST TRIPS -- a full trip of the state machine is followed by up to seven trips that only step (the pass was bound by vector
//     issue, two thirds of it the rare paths); GIANTS -- a lane long on its unit gives frames away before the queue is empty; units that verify no
//     candidates themselves (every small range inside a check window is a TASK record for ceKernel).
//
// Results are bit-identical to the CPU path: BGS-HostAlgnmtAlgo2.cpp:933-1071
// (order), :86-169 (check-and-extend), DV-Kernel.cu:437-955 (what the reference's own kernels do per step).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include "s3dp_device.h"
#include "s3dp_model.h"
#include "s3dp_search_common.h"

namespace s3 {

using namespace v2;

namespace unit {
enum : uint32_t { U_IDLE = 0, U_STEP = 1, U_LOAD = 2, U_ADOPT = 3, U_POPWAIT = 4, U_RESUME = 7, U_DONE = 8 };

}  // namespace unit

using namespace unit;

// COUNT: with the statistics of s3_batch_enable_counters (rank queries, trips, longest unit ...): ten more registers, measurement runs only
// (four waves per SIMD = 128 registers.  Five -- 96 registers, 29 dwords of the lane state spilled, 154 KB of LDS per CU -- was tried in round 5:
//  the unit passes of a ladder 149 -> 187 ms, profiles/README_r05.md)
// (Until round 5 a unit verified its first check-and-extend candidates itself -- two dependent lines per candidate inside a lane that moves in full trips only;
//  since the fast trips every candidate goes to ceKernel, and round 6 took the machinery out: S3DP_CE_GIVE_UP / S3DP_CE_INLINE_MAX are gone.)
template <bool UNIFORM, bool COUNT>
__global__ void __launch_bounds__(S3_SEARCH_BLOCK, S3_UNIT_WAVES_PER_SIMD)
unitKernel(DevIndex ix, SearchIn in, SearchOut out)
{
    extern __shared__ __align__(16) unsigned char shmem[];
    const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the wave number is uniform: said so, its addresses stay scalar)
    const uint32_t wpq = in.wpqLoad;                                        // words per orientation in readsT
    const uint32_t wl = in.wLds;                                            // words of the searched strand kept in LDS
    const uint32_t waveBytes = 4096u + S3_UNIT_RDY_BYTES + wl * 256u;
    unsigned char *wbase = shmem + S3_V2_MODEL_BYTES + wv * waveBytes;
    uint4 *stage = (uint4 *)wbase;                                          // [64 owners][4 x 16 B]; between trips: the mailboxes of an adoption
    uint32_t *rdy = (uint32_t *)(wbase + 4096u);                            // tested units waiting for a lane: rdy[field * 64 + k], k < rdyCount
    uint32_t *ro = (uint32_t *)(wbase + 4096u + S3_UNIT_RDY_BYTES);         // the unit's strand of its read, MSB first: ro[w * 64 + lane]
    uint32_t *myStage32 = (uint32_t *)(stage + lane * 4u);

    if (UNIFORM) {
        const uint32_t *src = (const uint32_t *)&in.models[in.uniformLen];
        for (uint32_t k = threadIdx.x; k < sizeof(S3Model) / 4; k += blockDim.x) ((uint32_t *)shmem)[k] = src[k];
        __syncthreads();
    }
    const uint32_t *padFilter = (const uint32_t *)(shmem + S3_V2_MODEL_BYTES + (S3_SEARCH_BLOCK / 64u) * waveBytes);      // behind the waves' parts (lookupPadded, s3dp_search_common.h)
    padFilterBuild(ix, (uint32_t *)(shmem + S3_V2_MODEL_BYTES + (S3_SEARCH_BLOCK / 64u) * waveBytes));

    const uint32_t groups = (uint32_t)in.nCases * 2u;
    const uint32_t nListed = *in.slotCount;
    const uint32_t U = in.unitsPerItem;
    const uint32_t nSlots = nListed * groups * U;                           // (< 2^31: s3dp_api.hip)
    uint4 *stk = out.unitStack + (size_t)(blockIdx.x * (S3_SEARCH_BLOCK / 64u) + wv) * S3_UNIT_STACK * 64u;     // frame d of lane k: stk[(d % S3_UNIT_STACK) * 64 + k]

    const unsigned long long tWave0 = COUNT ? wall_clock64() : 0ULL;        // (COUNT) the wave's own duration in 10 ns ticks, summed over the grid: how much of a launch is its tail
    // ---- statistics (COUNT) ----
    uint32_t nRank = 0, nCutSkip = 0, nCutMid = 0, unitTrips = 0, maxUnitTrips = 0, nAdopt = 0;
    unsigned long long laneTrips = 0, waveTrips = 0;
    uint32_t dStepLane = 0, dBookWave = 0, dPopLane = 0, dFastWave = 0, dFastLane = 0, dParked = 0;     // ... fast trips, their stepping lanes; lanes parked (U_RESUME / U_POPWAIT) at a full trip     // (COUNT) how often the rare paths run, and for how many lanes

    // ---- per-lane state ----
    uint32_t state = U_IDLE;
    // The range, seen from the direction of the step the lane is in: the ACTIVE range [aL, aR] is the one the step's rank queries move (the forward
    // range in a backward step, the reverse range in a forward step), the passive one ends at pR and has the same size.  Frames on the stack are
    // kept the same way (their step says which way round); the roles swap when a step of the other direction begins.
    uint32_t aL = 0, aR = 0, pR = 0;
    int s = 0, i = 0, ins = 0, occ = 0;
    StepC sc = {0, 0, 0, 0};
    int Li = UNIFORM ? (int)in.uniformLen : 0;
    uint32_t subN = 0;                                                      // item (case * 2 + strand) | number of steps of the case << 8 | padded lookup key << 16
    uint32_t listed = 0, unitInRead = 0;
    uint32_t cntIdx = 0, seq = 0, total = 0;
    // GROUPS (round 5).  A unit is the part of an item's search tree whose first mismatch sits in ONE GROUP of in.unitGroup consecutive positions of the first
    // mismatch step: the lane starts from the trunk's state at the group's first position and walks the trunk's own path -- no mismatch yet -- through the group,
    // taking the mismatch children at every position (children first, ascending, then the continuation: the CPU path's order) and DROPPING the continuation at
    // the group's last position, where the next unit begins.  On that path it is in the trunk's footsteps: where the trunk handed its range to check-and-extend
    // and ended, the unit ends too, silently -- the record is the trunk's.  (Until round 5 a group was one position: seven steps per unit with three and four
    // mismatches, and loading the read, repeating the trunk's step and retiring cost as much as the walk.)
    uint32_t groupEnd = 0;                                                  // first position behind the unit's group (0: the unit "none", behind the step's last position)
    uint32_t sp = 0, sb = 0;                                                // frames [sb, sp) of the ring are the lane's stack
    uint4 tos = make_uint4(0, 0, 0, 0);                                     // copy of frame sp - 1 ...
    bool tosValid = false;                                                  // ... when this is set
    uint32_t aux = 0;                                                       // U_LOAD: the read's number in readsT; U_ADOPT: where the frame lies
    uint32_t loadOff = 0;
    uint32_t pendMeta = 0, pendA = 0, pendB = 0;
    uint32_t pendC = 0, pendP = 0;                                          // of a task: the first suffix of its range in the REVERSE index; the read position next to the matched part | forward step << 16
                                                                            // (SearchOut::recAux: the read's 48 characters from there are cut out where the record is written, once per trip)
    uint32_t countPending = 0;
    uint32_t age = 0;                                                       // full trips the lane has spent on its unit (a GIANT: more than in.giantAge)
    // ---- wave-uniform ----
    uint32_t recCur = 0, recEnd = 0;                                        // reserved record block
    uint32_t nodeCur = 0, nodeEnd = 0;                                      // reserved sub-unit counters
    bool queueEmpty = false, basePending = false, noNodes = out.dynCap == 0u;
    uint32_t pendBase = 0;
    uint32_t rdyCount = 0, rdyTaken = 0;

#define casIdx (((subN) & 0xFFu) >> 1)
#define subIdx ((subN) & 0xFFu)
#define nSteps ((int)(((subN) >> 8) & 0xFFu))
#define itemPadded (((subN) >> 16) & 1u)                       /* the item's lookup key is a padded one (lookupPadded, s3dp_search_common.h): said in its task records */
    auto sUnit = [&]() -> uint32_t { return (in.sStarPacked >> (3u * casIdx)) & 7u; };     // the step the item is cut at (its first step that allows a mismatch)
    auto stepPtr = [&](int k) -> const uint32_t * {
        if constexpr (UNIFORM) return (const uint32_t *)&((const S3Model *)shmem)->cs[casIdx].st[k];
        else return (const uint32_t *)&in.models[Li].cs[casIdx].st[k];
    };
    auto readChar = [&](int pos) -> uint32_t { return (ro[((uint32_t)pos >> 4) * 64u + lane] >> (30u - 2u * ((uint32_t)pos & 15u))) & 3u; };
    auto oword = [&](uint32_t w) -> uint32_t { return w < wl ? ro[w * 64u + lane] : 0u; };
    auto readWindow = [&](int p0) -> uint32_t {               // 16 oriented characters from p0 >= 0, first one in the top bits
        const uint32_t w = (uint32_t)p0 >> 4, sh = ((uint32_t)p0 & 15u) * 2u;
        const uint32_t a = oword(w), b2 = oword(w + 1u);
        return sh ? (a << sh) | (b2 >> (32u - sh)) : a;
    };
    auto emit = [&](uint32_t kind, uint32_t a, uint32_t b, uint32_t nmis) {
        pendMeta = 0x80000000u | seq | (kind << 24) | (nmis << 25);
        pendA = a; pendB = b;
        seq++;
    };
    auto push = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t meta) {
        const uint4 f = make_uint4(a, b, c, meta);
        stk[(sp & (S3_UNIT_STACK - 1u)) * 64u + lane] = f;
        sp++;
        tos = f; tosValid = true;
    };
    auto loadStepCache = [&]() { sc = packStep(stepPtr(s)); };
    // first and last element of the FORWARD range (what records, tasks and the suffix array speak of); sc is the step the range was last moved by
    auto fwdL = [&]() -> uint32_t { return sc.dir() < 0 ? aL : pR - (aR - aL); };
    auto fwdR = [&]() -> uint32_t { return sc.dir() < 0 ? aR : pR; };

    // advance without memory until the lane needs ranks / SA / text / its next frame, or its unit is finished
    auto settle = [&](bool needPop) {
        for (;;) {
            if (total >= in.maxOut) { state = U_DONE; return; }
            if (needPop) {
                if (sp == sb) { state = U_DONE; return; }
                if (!tosValid) { state = U_POPWAIT; return; }               // the frame comes with the next trip's loads
                aL = tos.x; aR = tos.y; pR = tos.z;
                const uint32_t m = tos.w;
                s = (int)(m & 7u); i = (int)((m >> 3) & 0x7FFu); ins = (int)((m >> 14) & 7u); occ = (int)((m >> 17) & 7u);
                sp--; tosValid = false;
                needPop = false;
                if (s < nSteps) loadStepCache();
            }
            if (s >= nSteps) {                                               // SRA_STEP_TYPE_COMPLETE
                uint32_t l = fwdL(), rr_ = fwdR();
                if (aR >= aL) {
                    if (pendMeta >> 31) { state = U_RESUME; return; }        // one record per lane and trip: this frame waits for the next trip
                    if (total + (rr_ - l + 1u) > in.maxOut) rr_ = l + in.maxOut - total - 1u;
                    emit(0, l, rr_, (uint32_t)(ins + occ));
                    total += rr_ - l + 1u;
                    countPending += rr_ - l + 1u;
                }
                needPop = true;
                continue;
            }
            const bool exact = ins >= sc.maxErr();
            int lenEff = sc.len();
            if (!exact && sc.minErr() > ins + 1) lenEff -= sc.minErr() - ins - 1;
            const bool alive = aL <= aR;
            if (i < lenEff && alive) {
                const int pos = sc.start() + sc.dir() * i;
                if ((aR - aL) < (uint32_t)sc.ceThr() && pos >= sc.ceStart() && pos <= sc.ceEnd()) {
                    if (groupEnd != 0u && s == (int)sUnit() && ins == 0) { needPop = true; continue; }      // in the trunk's footsteps: the trunk wrote this task and ended here
                    if (pendMeta >> 31) { state = U_RESUME; return; }
                    pendMeta = 0x80000000u | seq | S3_REC_TASK | (sc.dir() > 0 ? S3_REC_FWD : 0u) | (itemPadded ? S3_REC_PADDED : 0u);
                    pendP = (uint32_t)pos | (sc.dir() > 0 ? 1u << 16 : 0u);
                    pendA = fwdL();
                    pendB = (aR - aL) | ((uint32_t)s | (uint32_t)i << 3 | (uint32_t)ins << 14 | (uint32_t)occ << 17) << 8;
                    pendC = aL;                                              // (forward step: the active range is the reverse index's)
                    seq += aR - aL + 1u;
                    needPop = true;
                    continue;
                }
                state = U_STEP;
                return;
            }
            if (ins >= sc.minErr() && alive) {
                occ += ins; ins = 0; s++; i = 0;
                if (s < nSteps) {
                    const int was = sc.dir();
                    loadStepCache();
                    if (sc.dir() != was) { const uint32_t oldR = aR; aL = pR - (aR - aL); aR = pR; pR = oldR; }      // the other index moves from here on
                }
                continue;
            }
            needPop = true;
        }
    };


    // ---- the STEP of a lane, in two halves around the trip's wait (the slow trips and the fast trips below share them) ----
    // what a stepping lane asks for: its two rank indices, the line of the left one (fetched by its quad) and -- when the right one lies in another line -- the
    // three pieces of that line it reads itself
    auto stepAsk = [&](uint32_t &iL, uint32_t &iR, bool &needB, int &dir, const uint4 *&ownLine, const uint4 *&aP0, const uint4 *&aP1, const uint32_t *&aD0) {
        dir = sc.dir();
        const uint32_t inv = dir < 0 ? ix.fwd.inverseSa0 : ix.rev.inverseSa0;
        iL = aL;
        iR = aR + 1u;
        iL -= (iL > inv);
        iR -= (iR > inv);
        const uint32_t kL = (iL + 63u) >> 7, kR = (iR + 63u) >> 7;
        ownLine = (dir < 0 ? ix.fwd.lines : ix.rev.lines) + (size_t)kL * 4u;
        needB = kR != kL;
        if (needB) {
            const uint4 *line = (dir < 0 ? ix.fwd.lines : ix.rev.lines) + (size_t)kR * 4u;
            const bool fw = iR > (kR << 7);
            aP0 = line;
            aP1 = line + (fw ? 2 : 1);
            aD0 = (const uint32_t *)(line + 3) + (fw ? 1 : 0);
        }
    };
    // the step itself, from the staged line: ranks, children, pushes.  Leaves doSettle / needPop for the lane's next move (doSettle false: the lane is done)
    auto stepTake = [&](uint32_t iL, uint32_t iR, bool needB, int dir, const uint4 &p0, const uint4 &p1, uint32_t d0, bool &doSettle, bool &needPop) {
        const bool exact = ins >= sc.maxErr();
        const RankSel qL = rankSel(iL), qR = rankSel(iR);
        const uint4 occA = stage[lane * 4u];
        // own line in LDS as words: [0..3] counts, [4..7] half before (reversed), [8..11] half after, [12] / [13] precounts
        const uint32_t offL = (qL.fwd ? 8u : 4u) + (qL.far ? 2u : 0u);
        const uint4 oL = rankFromWords(qL, occA, myStage32[offL], myStage32[offL + 1u], myStage32[12u + (qL.fwd ? 1u : 0u)]);
        // the right end's rank from ONE body of code whichever line it sits in: the operands are selected, not the results (two bodies in two branches ran one after the
        // other in nearly every trip)
        const uint32_t offR = (qR.fwd ? 8u : 4u) + (qR.far ? 2u : 0u);
        const uint4 occR = make_uint4(needB ? p0.x : occA.x, needB ? p0.y : occA.y, needB ? p0.z : occA.z, needB ? p0.w : occA.w);   // (by component: a select of two structures goes through memory)
        // (values first, then selects: a select between two members of a structure handed in by reference is a select of addresses, and the structure then lives in
        //  scratch memory)
        const uint32_t ownA = myStage32[offR], ownB = myStage32[offR + 1u], ownS = myStage32[12u + (qR.fwd ? 1u : 0u)];
        const uint32_t p1x = p1.x, p1y = p1.y, p1z = p1.z, p1w = p1.w;
        const uint32_t farA = qR.far ? p1z : p1x, farB = qR.far ? p1w : p1y;
        const uint32_t waR = needB ? farA : ownA, wbR = needB ? farB : ownB, subR = needB ? d0 : ownS;
        const uint4 oR = rankFromWords(qR, occR, waR, wbR, subR);
        if constexpr (COUNT) nRank += 2;
        const int pos = sc.start() + dir * i;
        const uint32_t ch = readChar(pos);
        // child ranges: the active range advances by rank, the passive one shrinks by the occurrences of the larger symbols
        const uint32_t e1 = oR.y - oL.y, e2 = oR.z - oL.z, e3 = oR.w - oL.w;
        const uint32_t cOc = occAbove(ch, e1, e2, e3);
        const uint32_t gAl = sel4(oL, ch) + 1u, gAr = sel4(oR, ch), gPr = pR - cOc;         // the continuation (the read's own character)
        doSettle = true;
        // on the trunk's path (no mismatch yet in the unit's step) at the group's last position, or at the step's last position for a first mismatch: only the
        // mismatch children -- the continuation belongs to the next unit, or to the unit "none"
        const int lenTrunk = sc.minErr() > 1 ? sc.len() - (sc.minErr() - 1) : sc.len();
        const bool branchOnly = groupEnd != 0u && s == (int)sUnit() && ins == 0 && ((uint32_t)i + 1u >= groupEnd || i + 1 >= lenTrunk);
        bool taken = false;                                                // the lane goes on in a mismatch child
        if (!exact) {
            const bool a0 = ch != 0 && oL.x + 1u <= oR.x, a1 = ch != 1 && oL.y + 1u <= oR.y;
            const bool a2 = ch != 2 && oL.z + 1u <= oR.z, a3 = ch != 3 && oL.w + 1u <= oR.w;
            if (a0 || a1 || a2 || a3) {
                if (sp - sb + 4u > S3_UNIT_STACK) { atomicOr(out.flags, S3_FLAG_STACK_OVERFLOW); state = U_DONE; sp = sb; tosValid = false; doSettle = false; }
                else {
                    // depth-first, children in ascending order, the continuation last: the lane goes on in the smallest child that is alive, the
                    // others wait on the stack (pushed in descending order), the continuation below them
                    const uint32_t mCont = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)ins << 14) | ((uint32_t)occ << 17);
                    const uint32_t mKid = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)(ins + 1) << 14) | ((uint32_t)occ << 17);
                    const uint32_t kid = a0 ? 0u : a1 ? 1u : a2 ? 2u : 3u;
                    if (!branchOnly) push(gAl, gAr, gPr, mCont);
                    if (a3 && kid != 3u) push(oL.w + 1u, oR.w, pR, mKid);
                    if (a2 && kid != 2u) push(oL.z + 1u, oR.z, pR - e3, mKid);
                    if (a1 && kid != 1u) push(oL.y + 1u, oR.y, pR - (e2 + e3), mKid);
                    const uint32_t kOc = occAbove(kid, e1, e2, e3);
                    aL = sel4(oL, kid) + 1u; aR = sel4(oR, kid); pR = pR - kOc;
                    ins++; i++;
                    taken = true;
                }
            } else if (branchOnly) needPop = true;                         // no child alive: the unit is empty
        } else if (branchOnly) needPop = true;
        if (doSettle && !needPop && !taken) { aL = gAl; aR = gAr; pR = gPr; i++; }
    };

    for (;;) {
        // ---------- retire finished units (their occurrences were booked at the end of the trip that finished them) ----------
        if (state == U_DONE) {
            if (seq != 0u || cntIdx >= out.dynBase) out.itemCounts[cntIdx] = seq;      // (a unit that wrote nothing leaves the counter alone: the trunk lane may own it)
            state = U_IDLE;
        }
        // ---------- idle lanes take tested units ----------
        unsigned long long idleMask = __ballot(state == U_IDLE);
        // (starting units is a divergent path of some two hundred instructions, and two or three lanes of a wave finish a unit on EVERY trip: the idle lanes
        //  wait until in.takeMin of them can start together -- or nothing else is running, or the queue is empty and waiting would only stretch the tail)
        if (idleMask && rdyTaken < rdyCount && ((uint32_t)__popcll(idleMask) >= in.takeMin || queueEmpty)) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
            const uint32_t e = rdyTaken + rank, avail = rdyCount - rdyTaken;
            if (state == U_IDLE && rank < avail) {
                aL = rdy[e]; aR = rdy[64u + e]; pR = rdy[128u + e];          // the trunk state {l, r, rev l}: turned the step's way when the step is known
                listed = rdy[192u + e]; aux = rdy[320u + e];
                const uint32_t sp_ = rdy[256u + e];                          // item | position << 8 (the division is made once per chunk, where the chunk is tested)
                const uint32_t sub = sp_ & 0xFFu, p = sp_ >> 8;
                unitInRead = sub * U + p;
                const uint32_t cas = sub >> 1;
                subN = sub | ((in.nStepsPacked >> (3u * cas)) & 7u) << 8;
                s = (int)((in.sStarPacked >> (3u * cas)) & 7u);
                cntIdx = in.unitBase + listed * groups * U + unitInRead;
                groupEnd = p + 1u != U ? (p + 1u) * in.unitGroup : 0u;
                i = (int)(p * in.unitGroup);                                                 // (the unit without a mismatch in this step: behind the step's last position, set when the step is known)
                ins = 0; occ = 0; seq = 0; total = 0; sp = 0; sb = 0; tosValid = false;
                countPending = 0; age = 0;
                if constexpr (COUNT) unitTrips = 0;
                loadOff = 0;
                state = U_LOAD;
            }
            const uint32_t nIdle = (uint32_t)__popcll(idleMask);
            rdyTaken += nIdle < avail ? nIdle : avail;
            idleMask = __ballot(state == U_IDLE);
        }
        // ---------- the queue: what this trip does for it ----------
        // GIANTS.  A unit is a few dozen steps; one in a million is ten thousand (the longest unit of the benchmark's phase 2: 13,000 trips of ONE lane, 60 ms --
        // the length of the whole launch), and until the queue was empty nobody helped it.  A lane that has been on its unit for more than in.giantAge full trips
        // and has frames to give is a giant: its wave takes no further chunks from the queue (the other waves take them), and as its lanes run out of units they
        // adopt the giant's oldest frames -- the mechanism of the tail, started while the rest of the device still has work (unit passes 115.9 -> 111.2 ms).
        // (Measured beside it and left out: records parked in LDS next to the one in registers, so that a lane with two or three task records need not wait
        //  a full trip per record -- 113.6 / 114.0 / 114.5 ms with 0 / 1 / 2 slots; trunk positions without a live mismatch child keeping no state: 111.0 / 111.2.)
        if (state != U_IDLE) age++;
        const bool canGive = state != U_IDLE && state != U_LOAD && state != U_ADOPT && sp > sb;
        const unsigned long long giantMask = in.giantAge && !noNodes ? __ballot(canGive && age > in.giantAge) : 0ULL;      // (no counters left for adopted frames: no giants)
        const bool tail = queueEmpty && !basePending && rdyTaken >= rdyCount;
        const bool early = !tail && giantMask && !basePending && rdyTaken >= rdyCount;
        const bool stageA = !queueEmpty && !basePending && !giantMask;      // take the next chunk's number (lane 0's atomic rides with this trip's loads)
        const bool stageB = basePending && rdyTaken >= rdyCount;            // load and test the chunk
        if ((tail || early) && idleMask) {
            if (tail && idleMask == ~0ULL) break;                           // nothing left anywhere
            // ---------- adoption: idle lanes take the oldest frames of busy lanes ----------
            // (in the tail only once half of the wave is idle: a lane that finishes a little before its neighbours is no tail, and every adopted frame takes a
            //  counter; from a giant: as soon as in.giantIdle lanes are free)
            // (in.tailAge: full trips a giver of the tail must have spent on its unit.  0: measured, round 5 -- with three and four mismatches a pass makes 8 M
            //  adoptions in 8 M wave trips, and asking the givers for 3 / 8 / 16 / 32 full trips first made the unit passes 111.3 / 112.5 / 114.7 / 121.0 ms (111.0))
            const bool gives = tail ? (canGive && age > in.tailAge) : (canGive && age > in.giantAge);
            const unsigned long long giveMask = tail ? __ballot(gives) : giantMask;
            uint32_t nPairs = (uint32_t)__popcll(giveMask);
            const uint32_t nIdle = (uint32_t)__popcll(idleMask);
            if (nIdle < nPairs) nPairs = nIdle;
            if (nIdle < (tail ? in.adoptIdle : in.giantIdle)) nPairs = 0;
            if (nPairs && !noNodes) {
                if (nodeCur + nPairs > nodeEnd) {
                    // (what is left of the old block is dropped: a few counters nobody uses)
                    uint32_t b0 = 0;
                    if (lane == 0) b0 = atomicAdd(out.dynUsed, S3_UNIT_NODE_BLOCK);
                    b0 = (uint32_t)__shfl((int)b0, 0);
                    nodeCur = b0 < out.dynCap ? b0 : out.dynCap;
                    nodeEnd = b0 + S3_UNIT_NODE_BLOCK < out.dynCap ? b0 + S3_UNIT_NODE_BLOCK : out.dynCap;
                    if (nodeEnd < nodeCur) nodeEnd = nodeCur;
                    if (nodeEnd == nodeCur) noNodes = true;                   // the counters are used up: from here on every lane finishes what it has
                }
                if (nodeEnd - nodeCur < nPairs) nPairs = nodeEnd - nodeCur;
            } else nPairs = 0;
            if (nPairs) {
                const uint32_t gRank = __builtin_amdgcn_mbcnt_hi((uint32_t)(giveMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)giveMask, 0u));
                const uint32_t iRank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
                if (gives && gRank < nPairs) {
                    const uint32_t node = nodeCur + gRank, newCnt = out.dynBase + node;
                    // the new counter follows the giver's: whatever the giver still finds comes before the frame it gives away.  (Only the lane that
                    // owns a counter writes its link, so it reads back what it wrote.)
                    uint32_t *myLink = out.chainNext + (cntIdx - in.unitBase);
                    const uint32_t after = *myLink;
                    out.chainNext[newCnt - in.unitBase] = after;
                    *myLink = newCnt;
                    out.dynOwner[node] = in.unitBase + listed * groups * U + unitInRead;
                    stage[gRank * 2u] = make_uint4(lane | (sb & (S3_UNIT_STACK - 1u)) << 8, listed, unitInRead, subN);
                    stage[gRank * 2u + 1u] = make_uint4(newCnt, (uint32_t)Li, groupEnd, 0u);     // (the frame may be the trunk path's continuation: the adopter must know where the group ends)
                    sb++;
                    if (sb == sp) tosValid = false;
                }
                __builtin_amdgcn_wave_barrier();
                if (state == U_IDLE && iRank < nPairs) {
                    const uint4 m0 = stage[iRank * 2u], m1 = stage[iRank * 2u + 1u];
                    const uint32_t giver = m0.x & 63u;
                    aux = ((m0.x >> 8) & (S3_UNIT_STACK - 1u)) * 64u + giver;
                    listed = m0.y; unitInRead = m0.z; subN = m0.w;
                    cntIdx = m1.x;
                    if constexpr (!UNIFORM) Li = (int)m1.y;
                    for (uint32_t w = 0; w < wl; w++) ro[w * 64u + lane] = ro[w * 64u + giver];
                    groupEnd = m1.z;
                    seq = 0; total = 0; sp = 0; sb = 0; tosValid = false;
                    countPending = 0; age = 0;
                    if constexpr (COUNT) { unitTrips = 0; nAdopt++; }
                    state = U_ADOPT;
                }
                nodeCur += nPairs;
                __builtin_amdgcn_wave_barrier();
            }
        }

        // ---------- what does this lane need?  Addresses only: every lane issues the same loads, back to back, with no branch between them (a load
        // inside a branch makes the compiler wait for it at the join); a lane that needs nothing reads a cache-hot dummy ----------
        uint32_t iL = 0, iR = 0;
        bool needB = false;
        int dir = 0;
        const uint4 *dummy16 = (const uint4 *)ix.pac;
        const uint4 *ownLine = ix.fwd.lines;                                    // the 64-byte line this lane wants fetched by its quad (nothing: line 0)
        const uint4 *aP0 = dummy16, *aP1 = dummy16, *aPT = dummy16;
        const uint32_t *aD0 = ix.pac, *aCut = ix.pac;
        if (state == U_STEP) stepAsk(iL, iR, needB, dir, ownLine, aP0, aP1, aD0);
        else if (state == U_LOAD) {
            const uint4 *src = (const uint4 *)(in.readsT + (size_t)aux * 2u * wpq + (size_t)(subN & 1u) * wpq + loadOff);
            aP0 = src;
            if (wpq - loadOff > 4u) aP1 = src + 1;
            if constexpr (!UNIFORM) aD0 = in.readLengths + aux;
        } else if (state == U_ADOPT) {
            aP0 = stk + aux;
        }
        const bool running = state == U_STEP || state == U_POPWAIT || state == U_RESUME;
        const bool wantTos = running && !tosValid && sp > sb;
        const bool wantCut = wantTos || state == U_ADOPT;
        if (wantTos) aPT = stk + ((sp - 1u) & (S3_UNIT_STACK - 1u)) * 64u + lane;
        if (wantCut) aCut = out.readCut + listed;
        // ---------- issue: the cooperative lines (4 adjacent lanes fetch the 4 x 16 B of one owner's line: the owner's address travels by ds_bpermute,
        // two words), then what single lanes need ----------
        const uint32_t q16 = lane & 3u, q4 = lane >> 2;
        const uint32_t ownLo = (uint32_t)(uintptr_t)ownLine, ownHi = (uint32_t)((uintptr_t)ownLine >> 32);
        auto quadLine = [&](uint32_t owner) -> uint4 {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownLo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownHi);
            return loadGlobal16(lo, hi, q16 * 16u);
        };
        const uint4 va0 = quadLine(q4), va1 = quadLine(16u + q4), va2 = quadLine(32u + q4), va3 = quadLine(48u + q4);
        const uint4 p0 = *aP0, p1 = *aP1, pT = *aPT;
        const uint32_t d0 = *aD0;
        const uint32_t dCut = __hip_atomic_load(aCut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the queue (wave-uniform branches; what they load is read under the same conditions further down)
        unsigned long long qBase = 0;
        uint4 cTs;
        uint32_t cCut, cRidx, cListed = 0, cRem = 0, rem0 = 0;
        bool cIn = false;
        if (stageA && lane == 0) qBase = atomicAdd(out.workHead, (unsigned long long)S3_UNIT_CHUNK);
        if (stageB) {
            const uint32_t slot = pendBase + lane;
            cIn = slot < nSlots;
            // (a chunk of 64 consecutive slots holds at most two values of slot / nListed: the division is made once, on the chunk's first slot, in scalar registers)
            const uint32_t base0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pendBase);
            const uint32_t slotC = cIn ? slot : base0;                      // (lanes past the end read what the chunk's first lane reads)
            rem0 = base0 / nListed;
            const uint32_t brk = (rem0 + 1u) * nListed;
            cRem = rem0 + (slotC >= brk ? 1u : 0u) + (nListed < 64u && slotC >= brk ? (slotC - brk) / nListed : 0u);
            cListed = slotC - cRem * nListed;
            cTs = out.trunkStates[slotC];                                    // tsAt(p) of (item, position) = slot / nListed, read = slot % nListed
            cCut = __hip_atomic_load(out.readCut + cListed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cRidx = out.listedRead[cListed];
        }
        // ---------- stage the cooperative lines in LDS (owner slot = 64 B) ----------
        stage[(q4) * 4u + q16] = va0;
        stage[(16u + q4) * 4u + q16] = va1;
        stage[(32u + q4) * 4u + q16] = va2;
        stage[(48u + q4) * 4u + q16] = va3;
        const uint32_t saVal = d0;
        __builtin_amdgcn_wave_barrier();

        // ---------- the queue's results ----------
        if (stageA) {
            const unsigned long long b = (unsigned long long)__shfl((long long)qBase, 0);
            if (b >= nSlots) queueEmpty = true;
            else { pendBase = (uint32_t)b; basePending = true; }
        }
        if (stageB) {
            // units with a trunk state that are not behind their read's cut, packed to the front of the wave's list
            const bool live = cIn && (cTs.w & S3_TRUNK_VALID) && !(cRem > cCut);
            const unsigned long long lm = __ballot(live);
            if (live) {
                const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                rdy[k] = cTs.x; rdy[64u + k] = cTs.y; rdy[128u + k] = cTs.z;
                const uint32_t sub0 = rem0 / U;                                  // (scalar: of the chunk's first slot) ...
                uint32_t cSub = sub0, cP = cRem - sub0 * U;                       // ... corrected per lane: cRem - rem0 is 0 or 1 unless the list is shorter than a chunk
                while (cP >= U) { cP -= U; cSub++; }
                rdy[192u + k] = cListed; rdy[256u + k] = cSub | cP << 8; rdy[320u + k] = cRidx;
            }
            rdyCount = (uint32_t)__popcll(lm); rdyTaken = 0;
            basePending = false;
            if constexpr (COUNT) nCutSkip += (cIn && (cTs.w & S3_TRUNK_VALID) && !live) ? 1u : 0u;
            __builtin_amdgcn_wave_barrier();
        }

        // ---------- advance ----------
        bool doSettle = false, needPop = false;
        if constexpr (COUNT) { waveTrips++; if (state != U_IDLE) { unitTrips++; laneTrips++; if (unitTrips > maxUnitTrips) maxUnitTrips = unitTrips; } }
        if (wantTos) { tos = pT; tosValid = true; }
        if (wantCut && unitInRead > dCut && state != U_ADOPT) {                  // behind the read's cut
            if constexpr (COUNT) nCutMid++;
            state = U_DONE; sp = sb; tosValid = false;
        }
        if (state == U_STEP) stepTake(iL, iR, needB, dir, p0, p1, d0, doSettle, needPop);
        else if (state == U_RESUME) { doSettle = true; needPop = false; }
        else if (state == U_POPWAIT) { doSettle = true; needPop = true; }
        else if (state == U_ADOPT) {
            aL = p0.x; aR = p0.y; pR = p0.z;
            s = (int)(p0.w & 7u); i = (int)((p0.w >> 3) & 0x7FFu); ins = (int)((p0.w >> 14) & 7u); occ = (int)((p0.w >> 17) & 7u);
            if (s < nSteps) loadStepCache();
            if (unitInRead > dCut) { state = U_DONE; if constexpr (COUNT) nCutMid++; }
            else { doSettle = true; needPop = false; }
        } else if (state == U_LOAD) {
            const uint32_t vals[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t w = loadOff + k;
                if (w < wl) ro[w * 64u + lane] = vals[k];
            }
            loadOff += 8;
            if constexpr (!UNIFORM) Li = (int)saVal;
            if (loadOff >= wl || loadOff >= wpq) {
                {
                    const uint32_t *l0 = stepPtr(0);                                  // the case's lookup step: start (low half of word 0), len (high half of word 2), type (byte 1 of word 3)
                    if (lookupPadded(ix, padFilter, (int)(int8_t)((l0[3] >> 8) & 0xFFu), (int)(int16_t)(l0[0] & 0xFFFFu), (int)l0[2] >> 16, readWindow)) subN |= 1u << 16;
                }
                loadStepCache();
                {
                    // the trunk state {l, r, rev l} the step's way round
                    const uint32_t l = aL, r = aR, rvl = pR;
                    if (sc.dir() < 0) pR = rvl + (r - l);
                    else { aL = rvl; aR = rvl + (r - l); pR = r; }
                }
                if (groupEnd) state = U_STEP;
                else { i = sc.len(); doSettle = true; needPop = false; }         // the unit without a mismatch in this step: behind its last position
            }
        }
        if constexpr (COUNT) {
            dStepLane += state == U_STEP ? 1u : 0u;
            dPopLane += (doSettle && needPop) || state == U_POPWAIT ? 1u : 0u;
            dParked += state == U_RESUME || state == U_POPWAIT ? 1u : 0u;
        }
        if (doSettle) settle(needPop);
        // ---------- book the occurrences the unit holds for sure, where the other units of its read can see them: per item (itemAcc) and per position
        // (the spare word of the trunk states), S3_V2_COUNT_BATCH at a time and when the unit ends.  Everything in sequence order up to and including this
        // unit that adds up to MaxOutputPerRead puts the read's cut behind this unit; the items up to this one adding up to it put it behind the item
        // (s3dp_search_common.h, "the cut") ----------
        if constexpr (COUNT) dBookWave += __ballot(countPending >= S3_V2_COUNT_BATCH || (state == U_DONE && countPending)) ? 1u : 0u;
        if (countPending >= S3_V2_COUNT_BATCH || (state == U_DONE && countPending)) {
            const uint32_t cnt = countPending, sub = subIdx;
            countPending = 0;
            uint32_t *cutPtr = out.readCut + listed;
            uint32_t *acc = out.itemAcc + (size_t)listed * groups;
            atomicAdd(&acc[sub], cnt);
            uint32_t before = 0;
            for (uint32_t q = 0; q < sub; q++) before += __hip_atomic_load(&acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t mine = __hip_atomic_load(&acc[sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + mine >= in.maxOut) atomicMin(cutPtr, sub * U + U - 1u);
            const uint32_t myPos = unitInRead - sub * U;
            uint4 *ts0 = out.trunkStates + (size_t)sub * U * nListed + listed;                 // position q of this item and read: ts0[q * nListed]
            atomicAdd(&ts0[(size_t)myPos * nListed].w, cnt);
            uint32_t upTo = before;
            for (uint32_t q = 0; q <= myPos; q++) upTo += __hip_atomic_load(&ts0[(size_t)q * nListed].w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~S3_TRUNK_VALID;
            if (upTo >= in.maxOut) atomicMin(cutPtr, unitInRead);
        }
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
                    if (recCur + rk < recEnd) {
                        out.recs[recCur + rk] = make_uint4(cntIdx | S3_REC_PASS2, pendMeta & 0x7FFFFFFFu, pendA, pendB);
                        if (pendMeta & S3_REC_TASK) {
                            uint32_t q0, q1, q2;
                            adjacentWindows((int)(pendP & 0xFFFFu), (pendP >> 16) ? 1 : -1, oword, q0, q1, q2);
                            out.recAux[recCur + rk] = make_uint4(pendC, q0, q1, q2);
                        }
                    }
                    pendMeta = 0;
                }
                recCur += np;
                if (recCur > recEnd) recCur = recEnd;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---------- FAST TRIPS (round 5).  The trip above carries every path of the state machine, and with 64 lanes some lane needs every one of them on
        // nearly every trip: ~890 vector instructions per trip at 26 active lanes of 64 (profiles/README_r05.md), of which the step itself is a third -- the
        // pass is bound by vector issue, not by its lines.  So while at least in.fastMin lanes of the wave are stepping, up to in.fastTrips trips that do
        // NOTHING BUT STEP follow: the stepping lanes' lines, the one wait, ranks, children, pushes, and the test whether the lane's next move is a
        // step again.  A lane whose next move is anything else -- pop, record, step boundary, check-and-extend, end of the unit -- parks in U_RESUME /
        // U_POPWAIT and makes it in the next full trip, together with the other lanes parked meanwhile: the rare paths run for many lanes at a time
        // instead of two or three.  A lane is nobody else's business (the cut and the booking are order-free), so the records do not change. ----------
        // (What a fast trip carries is what it costs: its ~330 vector instructions are half of what the four waves of a SIMD can issue inside one memory
        //  latency.  Round 5 measured three richer variants, all bit-exact: the reads of newly taken units loaded in the fast trips with retirement at the end
        //  of the full trip (unit passes of a ladder 128.3 -> 130.4 ms), settle() and the frame under a popped one in the fast trips (parked lane trips
        //  250 M -> 67 M, wave trips -11 %, but 3.9 -> 5.3 us per fast trip: 155 ms), fast trips for any stepping lane once the queue is empty (+11 ms:
        //  frames are adopted in full trips only).  profiles/README_r05.md.)
        for (uint32_t ft = 0; ft < in.fastTrips; ft++) {
            if ((uint32_t)__popcll(__ballot(state == U_STEP)) < in.fastMin) break;
            uint32_t iL = 0, iR = 0;
            bool needB = false;
            int dir = 0;
            const uint4 *ownLine = ix.fwd.lines;
            const uint4 *aP0 = (const uint4 *)ix.pac, *aP1 = (const uint4 *)ix.pac;
            const uint32_t *aD0 = ix.pac;
            if (state == U_STEP) stepAsk(iL, iR, needB, dir, ownLine, aP0, aP1, aD0);
            const uint32_t q16 = lane & 3u, q4 = lane >> 2;
            const uint32_t ownLo = (uint32_t)(uintptr_t)ownLine, ownHi = (uint32_t)((uintptr_t)ownLine >> 32);
            auto quadLine = [&](uint32_t owner) -> uint4 {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownLo);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownHi);
                return loadGlobal16(lo, hi, q16 * 16u);
            };
            const uint4 va0 = quadLine(q4), va1 = quadLine(16u + q4), va2 = quadLine(32u + q4), va3 = quadLine(48u + q4);
            const uint4 p0 = *aP0, p1 = *aP1;
            const uint32_t d0 = *aD0;
            stage[(q4) * 4u + q16] = va0;
            stage[(16u + q4) * 4u + q16] = va1;
            stage[(32u + q4) * 4u + q16] = va2;
            stage[(48u + q4) * 4u + q16] = va3;
            __builtin_amdgcn_wave_barrier();
            if constexpr (COUNT) { waveTrips++; dFastWave++; if (state != U_IDLE) { unitTrips++; laneTrips++; if (unitTrips > maxUnitTrips) maxUnitTrips = unitTrips; } dStepLane += state == U_STEP ? 1u : 0u; dFastLane += state == U_STEP ? 1u : 0u; }
            if (state == U_STEP) {
                bool doSettle = false, needPop = false;
                stepTake(iL, iR, needB, dir, p0, p1, d0, doSettle, needPop);
                if (doSettle) {
                    // settle()'s way back to U_STEP, and nothing else of it
                    bool go = false;
                    if (!needPop) {
                        const bool exact = ins >= sc.maxErr();
                        int lenEff = sc.len();
                        if (!exact && sc.minErr() > ins + 1) lenEff -= sc.minErr() - ins - 1;
                        if (i < lenEff && aL <= aR) {
                            const int pos = sc.start() + sc.dir() * i;
                            go = !((aR - aL) < (uint32_t)sc.ceThr() && pos >= sc.ceStart() && pos <= sc.ceEnd());
                        }
                    }
                    state = go ? U_STEP : needPop ? U_POPWAIT : U_RESUME;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef casIdx
#undef subIdx
#undef nSteps
    for (uint32_t k = recCur + lane; k < recEnd; k += 64u) out.recs[k] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if constexpr (COUNT) {
        if (out.counters) {
            atomicAdd(&out.counters[0], (unsigned long long)nRank);
            atomicAdd(&out.counters[4], (unsigned long long)nCutSkip);
            atomicAdd(&out.counters[5], (unsigned long long)nCutMid);
            atomicAdd(&out.counters[6], laneTrips);
            atomicMax(&out.counters[7], (unsigned long long)maxUnitTrips);
            atomicAdd(&out.counters[8], (unsigned long long)nAdopt);
            if (lane == 0) atomicAdd(&out.counters[9], waveTrips);
            atomicAdd(&out.counters[13], (unsigned long long)dStepLane);
            if (lane == 0) atomicAdd(&out.counters[14], (unsigned long long)dBookWave);
            atomicAdd(&out.counters[15], (unsigned long long)dPopLane);
            if (lane == 0) atomicAdd(&out.counters[16], (unsigned long long)dFastWave);
            atomicAdd(&out.counters[17], (unsigned long long)dFastLane);
            atomicAdd(&out.counters[18], (unsigned long long)dParked);
            if (lane == 0) { atomicAdd(&out.counters[19], wall_clock64() - tWave0); atomicAdd(&out.counters[20], 1ULL); }
        }
    }
}

// per listed read: its index in readsT / readLengths (what the first pass had in its `active` indirection)
__global__ void listedReadsKernel(const uint32_t *heavyList, const uint32_t *heavyCount, const uint32_t *active, uint32_t *listedRead)
{
    const uint32_t n = *heavyCount;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t slot = heavyList[k];
        listedRead[k] = active ? active[slot] : slot;
    }
}

// LOCALITY (round 5).  The first pass lists heavy reads in the order its lanes give up on them: neighbours in the list have nothing in common, so the 64
// units a wave takes -- consecutive listed reads at one (item, position) -- walk 64 unrelated subtrees: every rank query its own line from HBM, every lane
// in its own state.  Reads out of the same place of the same repeat family walk nearly the same subtree.  So the list is put in the order of the reads'
// first 32 characters before the trunk pass: family-mates become neighbours, ask for the same rank lines in the same trip (one request to L2 instead of
// sixty-four to HBM) and branch the same way.  Units are numbered by (listed read, item, position) and every consumer finds a read through heavyIndex:
// the order of the list is nobody's business, the records come out where they came out before.
__global__ void listedKeysKernel(const uint32_t *slots, uint32_t n, const uint32_t *active, const uint32_t *readsT, uint32_t stride, uint32_t words,
                                 unsigned long long *keys, uint32_t *vals)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t slot = slots ? slots[k] : k;
        const uint32_t r = active ? active[slot] : slot;
        const uint32_t *w = readsT + (size_t)r * stride;
        keys[k] = (unsigned long long)w[0] << 32 | (words > 1u ? w[1] : 0u);
        vals[k] = slot;
    }
}

__global__ void listedPutKernel(const uint32_t *sortedSlots, uint32_t n, uint32_t *heavyList, uint32_t *heavyIndex)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t slot = sortedSlots[k];
        heavyList[k] = slot;
        heavyIndex[slot] = k;
    }
}

// The counters of adopted frames, chained behind the units they came from: a unit's count becomes the total of its chain, every node gets its place
// inside that total (what comes before it in sequence order).  One lane per unit with a chain; chains are short.
__global__ void unitChainKernel(SearchOut out, uint32_t unitBase, uint32_t nUnits)
{
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nUnits; u += gridDim.x * blockDim.x) {
        uint32_t node = out.chainNext[u];
        if (!node) continue;
        uint32_t off = out.itemCounts[unitBase + u];
        while (node) {
            out.dynLocal[node - out.dynBase] = off;
            off += out.itemCounts[node];
            node = out.chainNext[node - unitBase];
        }
        out.itemCounts[unitBase + u] = off;
    }
}

unsigned unitPassMaxWaves(unsigned maxBlocks) { return maxBlocks * (S3_SEARCH_BLOCK / 64u); }

int launchUnitPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s)
{
    if (in.nActive == 0) return (int)hipSuccess;
    uint64_t blocks = ((uint64_t)in.nActive * (uint64_t)in.nCases * 2u * in.unitsPerItem + S3_SEARCH_BLOCK - 1) / S3_SEARCH_BLOCK;
    if (blocks > maxBlocks) blocks = maxBlocks;
    const size_t lds = S3_V2_MODEL_BYTES + (size_t)(S3_SEARCH_BLOCK / 64) * (4096u + S3_UNIT_RDY_BYTES + in.wLds * 256u) + v2::S3_PAD_FILTER_BYTES;
    const bool count = out.counters != nullptr;
    using Kern = void (*)(DevIndex, SearchIn, SearchOut);
    static const Kern kerns[4] = {unitKernel<false, false>, unitKernel<false, true>, unitKernel<true, false>, unitKernel<true, true>};
    const Kern fn = kerns[(in.uniformLen ? 2 : 0) + (count ? 1 : 0)];
    if (lds > 48u * 1024u) {
        const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(S3_SEARCH_BLOCK), lds, s, ix, in, out);
    return (int)hipSuccess;
}

void launchListedReads(const uint32_t *heavyList, const uint32_t *heavyCount, const uint32_t *active, uint32_t *listedRead, uint32_t nHeavy, hipStream_t s)
{
    if (!nHeavy) return;
    hipLaunchKernelGGL(listedReadsKernel, dim3((nHeavy + 255u) / 256u), dim3(256), 0, s, heavyList, heavyCount, active, listedRead);
}

void launchListedKeys(const uint32_t *slots, uint32_t n, const uint32_t *active, const uint32_t *readsT, uint32_t stride, uint32_t words, unsigned long long *keys,
                      uint32_t *vals, hipStream_t s)
{
    if (!n) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)n + 255u) / 256u, 8192u);
    hipLaunchKernelGGL(listedKeysKernel, dim3(blocks), dim3(256), 0, s, slots, n, active, readsT, stride, words, keys, vals);
}

void launchListedPlace(const uint32_t *sortedSlots, uint32_t n, uint32_t *heavyList, uint32_t *heavyIndex, hipStream_t s)
{
    if (!n) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)n + 255u) / 256u, 8192u);
    hipLaunchKernelGGL(listedPutKernel, dim3(blocks), dim3(256), 0, s, sortedSlots, n, heavyList, heavyIndex);
}

void launchUnitChains(const SearchOut &out, uint32_t unitBase, uint32_t nUnits, hipStream_t s)
{
    if (!nUnits || !out.dynCap) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)nUnits + 255u) / 256u, 8192u);
    hipLaunchKernelGGL(unitChainKernel, dim3(blocks), dim3(256), 0, s, out, unitBase, nUnits);
}

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadUnit() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::unitKernel<true, false>); (void)hipGetLastError(); }
