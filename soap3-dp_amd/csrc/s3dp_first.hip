// s3dp_first.hip -- the FIRST PASS of the 2BWT k-mismatch search for gfx950: one lane per read, every (case, strand) item of the read in the CPU
// path's order (BGS-HostAlgnmtAlgo2.cpp:933-1071: case 0 '+', case 0 '-', case 1 '+' ...).
//
// The machine is the one of the unit pass (s3dp_unit.hip): persistent waves, a flat state machine per lane, ONE wait per trip behind loads that
// every lane issues without a branch in between, rank lines fetched by quads (the owner's address travels by ds_bpermute) and staged in LDS, the
// model step of a lane in four registers, the range kept the way round its step moves it (active / passive), the depth-first stack a ring in
// global memory with its top frame in registers.  What the first pass adds:
//   * the read is loaded in both orientations (two trips for 100 bases) and its items run back to back in the lane;
//   * an item starts from the longest lookup table the index holds: the 13-mer tables of the file format (with the range reconciliation of
//     BWTExactModelForward_Lookup2, BGS-HostAlgnmtAlgo2.cpp:349-356) or the 14- / 15- / 16-mer tables built in HBM at upload;
//   * check-and-extend is not done here: a small range inside the check window becomes a TASK record for ceKernel (s3dp_ce.hip) with one
//     sequence number reserved per candidate; a single suffix left inside the exact lookup step is handed over at once (S3_REC_EARLY);
//   * a read that reserves more candidates than in.heavyLimit, or costs more rank queries than in.heavyRank, is a HEAVY read: the lane lists it
//     for the second pass (trunk pass + unit pass) and drops it.
// The whole story of the search is told in s3dp_search_common.h.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include "s3dp_device.h"
#include "s3dp_model.h"
#include "s3dp_search_common.h"

namespace s3 {

using namespace v2;

namespace first {
enum : uint32_t { F_IDLE = 0, F_STEP = 1, F_LOAD = 2, F_KEYS = 3, F_LT = 4, F_POPWAIT = 5, F_RESUME = 6, F_DONE = 7 };
}
using namespace first;

// TRUNK: the trunk pass over the listed reads (in.slotCount of them): a slot is ONE (listed read, item); the lane walks the item's trunk -- lookup step,
// exact steps, then the first step that allows a mismatch WITHOUT taking one -- and leaves the range it holds at every position of that step, and
// behind its last one, in out.trunkStates: the unit pass starts from them.  Whatever ends the trunk early (a range handed to check-and-extend, a dead
// range) ends it for every unit: positions it never reached keep no state.  Its records (tasks for ceKernel) count in the counter of the item's last
// unit.  With in.unitsPerItem == 1 (no mismatch step to cut the item at: exact-match models; or S3DP_UNITS=0) the lane searches the whole item.
template <bool UNIFORM, bool COUNT, bool TRUNK>
// (the trunk role carries a few registers more than the first pass and is a thirtieth of a ladder's time: it gets the budget of three waves per SIMD instead of
//  spilling three of them into scratch memory)
__global__ void __launch_bounds__(S3_SEARCH_BLOCK, TRUNK ? S3_FIRST_WAVES_PER_SIMD - 1 : S3_FIRST_WAVES_PER_SIMD)
firstKernel(DevIndex ix, SearchIn in, SearchOut out)
{
    extern __shared__ __align__(16) unsigned char shmem[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t wpq = in.wpqLoad;                                        // words per orientation in readsT
    const uint32_t wl = in.wLds;                                            // words per orientation kept in LDS
    const uint32_t waveBytes = 4096u + S3_FIRST_RDY_BYTES + 2u * wl * 256u;
    unsigned char *wbase = shmem + S3_V2_MODEL_BYTES + wv * waveBytes;
    uint4 *stage = (uint4 *)wbase;                                          // [64 owners][4 x 16 B]
    uint32_t *rdy = (uint32_t *)(wbase + 4096u);                            // reads waiting for a lane: rdy[k] = slot, rdy[64 + k] = read
    uint32_t *ro = (uint32_t *)(wbase + 4096u + S3_FIRST_RDY_BYTES);        // both orientations of the read, MSB first: ro[(strand * wl + w) * 64 + lane]
    uint32_t *myStage32 = (uint32_t *)(stage + lane * 4u);

    if (UNIFORM) {
        const uint32_t *src = (const uint32_t *)&in.models[in.uniformLen];
        for (uint32_t k = threadIdx.x; k < sizeof(S3Model) / 4; k += blockDim.x) ((uint32_t *)shmem)[k] = src[k];
        __syncthreads();
    }
    const uint32_t *padFilter = (const uint32_t *)(shmem + S3_V2_MODEL_BYTES + (S3_SEARCH_BLOCK / 64u) * waveBytes);      // behind the waves' parts (lookupPadded, s3dp_search_common.h)
    padFilterBuild(ix, (uint32_t *)(shmem + S3_V2_MODEL_BYTES + (S3_SEARCH_BLOCK / 64u) * waveBytes));

    const uint32_t groups = (uint32_t)in.nCases * 2u;
    const uint32_t nListed = TRUNK ? *in.slotCount : 0u;
    const uint32_t U = TRUNK ? in.unitsPerItem : 1u;
    const bool onTrunk = TRUNK && U > 1u;                                   // the item is cut into units: the lane stays on the trunk
    const uint32_t nSlots = TRUNK ? nListed * groups : in.nActive;
    // (TRUNK) the lane's read in the list and the first step of its case that allows a mismatch are worked out where they are used (`listed`, `sStar` below):
    // as two more registers per lane they were what pushed the trunk role of this kernel into scratch memory -- 16 bytes per lane, and with them the
    // reallocation of the device's scratch buffer at the first launch on a stream (s3dp_api.hip, ListedSortConfig)
    uint4 *stk = out.unitStack + (size_t)(blockIdx.x * (S3_SEARCH_BLOCK / 64u) + wv) * S3_UNIT_STACK * 64u;     // frame d of lane k: stk[(d % S3_UNIT_STACK) * 64 + k]

    const unsigned long long tWave0 = COUNT ? wall_clock64() : 0ULL;        // (COUNT) the wave's own duration in 10 ns ticks
    unsigned long long nListedRank = 0;                                     // (COUNT) rank queries spent on reads that were then listed: thrown away, the second pass starts over
    uint32_t nRank = 0, nLt = 0, nWaveTrips = 0, nLaneTrips = 0, nStepLane = 0, nFastWave = 0, nFastLane = 0, nParked = 0;     // (COUNT)

    // ---- per-lane state ----
    uint32_t state = F_IDLE;
    uint32_t aL = 0, aR = 0, pR = 0;                                        // the range, the way round its step moves it (see s3dp_unit.hip)
    int s = 0, i = 0, ins = 0, occ = 0;
    StepC sc = {0, 0, 0, 0};
    int Li = UNIFORM ? (int)in.uniformLen : 0;
    uint32_t subN = 0;                                                      // item (case * 2 + strand) | number of steps of the case << 8
    uint32_t slotIdx = 0, rIdx = 0, cntIdx = 0, seq = 0, total = 0;
    uint32_t readTotal = 0;                                                 // occurrences the read's finished items have for sure
    uint32_t readCand = 0, readRank = 0;                                    // candidates reserved / rank queries spent by this read
    bool goHeavy = false;
    uint32_t sp = 0;                                                        // frames [0, sp) of the ring are the lane's stack
    uint4 tos = make_uint4(0, 0, 0, 0);
    bool tosValid = false;
    uint32_t loadOff = 0;
    uint32_t ltA = 0, ltB = 0, ltC = 0, ltD = 0;                            // lookup-table keys of the starting item
    bool padded = false;                                                    // the item's lookup range may hold an element that is no occurrence
    uint32_t pendMeta = 0, pendA = 0, pendB = 0;
    uint32_t pendC = 0, pendP = 0;                                          // of a task: the first suffix of its range in the REVERSE index; the read position next to the matched part | forward step << 16
                                                                            // (SearchOut::recAux: the read's 48 characters from there are cut out where the record is written, once per trip)
    // ---- wave-uniform ----
    uint32_t recCur = 0, recEnd = 0;
    bool queueEmpty = false, basePending = false;
    uint32_t pendBase = 0, rdyCount = 0, rdyTaken = 0;

#define listed (slotIdx - ((subN) & 0xFFu) * nListed)
#define sStar ((int)((in.sStarPacked >> (3u * (((subN) & 0xFFu) >> 1))) & 7u))
#define casIdx (((subN) & 0xFFu) >> 1)
#define subIdx ((subN) & 0xFFu)
#define strand ((subN) & 1u)
#define nSteps ((int)((subN) >> 8))
    auto stepPtr = [&](int k) -> const uint32_t * {
        if constexpr (UNIFORM) return (const uint32_t *)&((const S3Model *)shmem)->cs[casIdx].st[k];
        else return (const uint32_t *)&in.models[Li].cs[casIdx].st[k];
    };
    auto stepAt = [&](int k) -> S3Step {
        if constexpr (UNIFORM) return ((const S3Model *)shmem)->cs[casIdx].st[k];
        else return in.models[Li].cs[casIdx].st[k];
    };
    auto oword = [&](uint32_t w) -> uint32_t { return w < wl ? ro[(strand * wl + w) * 64u + lane] : 0u; };
    auto readChar = [&](int pos) -> uint32_t { return (ro[(strand * wl + ((uint32_t)pos >> 4)) * 64u + lane] >> (30u - 2u * ((uint32_t)pos & 15u))) & 3u; };
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
    auto fwdL = [&]() -> uint32_t { return sc.dir() < 0 ? aL : pR - (aR - aL); };
    auto fwdR = [&]() -> uint32_t { return sc.dir() < 0 ? aR : pR; };
    // (TRUNK) the range at position q of the mismatch step, as {l, r, rev l}
    auto leaveState = [&](uint32_t q) {
        const uint32_t size = aR - aL;
        const uint32_t fl = sc.dir() < 0 ? aL : pR - size, fr = sc.dir() < 0 ? aR : pR, rl = sc.dir() < 0 ? pR - size : aL;
        out.trunkStates[((size_t)subIdx * U + q) * nListed + listed] = make_uint4(fl, fr, rl, S3_TRUNK_VALID);
    };
    // a small range inside the check window, or a single suffix inside the lookup step: one task record, a sequence number per candidate
    auto task = [&](uint32_t flags) {
        pendMeta = 0x80000000u | seq | S3_REC_TASK | flags | (sc.dir() > 0 ? S3_REC_FWD : 0u) | (padded ? S3_REC_PADDED : 0u);
        pendP = (uint32_t)(sc.start() + sc.dir() * i) | (sc.dir() > 0 ? 1u << 16 : 0u);
        pendA = fwdL();
        pendB = (aR - aL) | ((uint32_t)s | (uint32_t)i << 3 | (uint32_t)ins << 14 | (uint32_t)occ << 17) << 8;
        pendC = aL;                                                          // (forward step: the active range is the reverse index's)
        seq += aR - aL + 1u;
        readCand += aR - aL + 1u;
    };

    // advance without memory until the lane needs ranks / its next frame or its item is finished
    auto settle = [&](bool needPop) {
        for (;;) {
            if (readTotal + total >= in.maxOut) { state = F_DONE; return; }
            if (needPop) {
                if (sp == 0u) { state = F_DONE; return; }
                if (!tosValid) { state = F_POPWAIT; return; }
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
                    if (pendMeta >> 31) { state = F_RESUME; return; }        // one record per lane and trip
                    if (readTotal + total + (rr_ - l + 1u) > in.maxOut) rr_ = l + in.maxOut - readTotal - total - 1u;
                    emit(0, l, rr_, (uint32_t)(ins + occ));
                    total += rr_ - l + 1u;
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
                    if (pendMeta >> 31) { state = F_RESUME; return; }
                    task(0u);
                    if (!TRUNK && readCand > in.heavyLimit) { goHeavy = true; state = F_DONE; return; }
                    needPop = true;
                    continue;
                }
                // Few suffixes left inside the exact lookup step: walking the rest of the step can only keep or lose each of them, in the same relative
                // order, and the next step hands them to check-and-extend at its first position anyway.  Hand them over now; the task also verifies what is
                // left of this step, inside the text (s3dp_search_common.h tells the whole story).  Not for ranges that may hold a padded table element.
                if (in.earlyCe && s == 0 && i >= (int)in.earlyMinI && sc.ceThr() == 0 && nSteps > 1 && !padded) {
                    const S3Step nx = stepAt(1);
                    const uint32_t lim = in.earlyCe < (uint32_t)nx.ceThr ? in.earlyCe : (uint32_t)nx.ceThr;
                    const int nxLenEff = nx.minErr > 1 ? nx.len - (nx.minErr - 1) : nx.len;
                    if ((aR - aL) < lim && nxLenEff > 0 && nx.start >= nx.ceStart && nx.start <= nx.ceEnd) {
                        if (pendMeta >> 31) { state = F_RESUME; return; }
                        task(S3_REC_EARLY);
                        if (!TRUNK && readCand > in.heavyLimit) { goHeavy = true; state = F_DONE; return; }
                        needPop = true;
                        continue;
                    }
                }
                state = F_STEP;
                return;
            }
            if (ins >= sc.minErr() && alive) {
                if (onTrunk && s == sStar) {
                    // past the last position of the mismatch step without a mismatch: the item's last unit ("none") goes on from here, in its turn
                    leaveState(U - 1u);
                    state = F_DONE;
                    return;
                }
                occ += ins; ins = 0; s++; i = 0;
                if (s < nSteps) {
                    const int was = sc.dir();
                    loadStepCache();
                    if (sc.dir() != was) { const uint32_t oldR = aR; aL = pR - (aR - aL); aR = pR; pR = oldR; }
                }
                continue;
            }
            needPop = true;
        }
    };


    // ---- the STEP of a lane, in two halves around the trip's wait (the full trips and the fast trips below share them; s3dp_unit.hip tells why) ----
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
    auto stepTake = [&](uint32_t iL, uint32_t iR, bool needB, int dir, const uint4 &p0, const uint4 &p1, uint32_t d0, bool &doSettle, bool &needPop) {
        const bool exact = ins >= sc.maxErr();
        const RankSel qL = rankSel(iL), qR = rankSel(iR);
        const uint4 occA = stage[lane * 4u];
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
        readRank += 2;
        const int pos = sc.start() + dir * i;
        const uint32_t ch = readChar(pos);
        const uint32_t e1 = oR.y - oL.y, e2 = oR.z - oL.z, e3 = oR.w - oL.w;
        const uint32_t cOc = occAbove(ch, e1, e2, e3);
        const uint32_t gAl = sel4(oL, ch) + 1u, gAr = sel4(oR, ch), gPr = pR - cOc;         // the continuation (the read's own character)
        doSettle = true;
        bool taken = false;
        const bool trunkHere = onTrunk && s == sStar;                      // no mismatch is taken in this step: the units take them
        if (trunkHere && in.unitGroup > 1u) {
            // a unit is a group of in.unitGroup positions: the state is left where a group begins (the unit walks the trunk's path through its group itself)
            const uint32_t g = (uint32_t)i / in.unitGroup;
            if (g + 1u < U && (uint32_t)i == g * in.unitGroup) leaveState(g);
        } else if (trunkHere && (uint32_t)i + 1u < U) {
            // A position none of whose mismatch children is alive keeps no state: its unit would load its read, make this very step, find nothing and retire --
            // five trips of a lane for nothing, and with three and four mismatches most units are of that kind (round 5).  The ranks are the ones the unit would
            // compute: same range, same position, same character.
            const bool anyKid = (ch != 0 && oL.x + 1u <= oR.x) || (ch != 1 && oL.y + 1u <= oR.y) || (ch != 2 && oL.z + 1u <= oR.z) || (ch != 3 && oL.w + 1u <= oR.w);
            if (anyKid) leaveState((uint32_t)i);
        }
        if (!exact && !trunkHere) {
            const bool a0 = ch != 0 && oL.x + 1u <= oR.x, a1 = ch != 1 && oL.y + 1u <= oR.y;
            const bool a2 = ch != 2 && oL.z + 1u <= oR.z, a3 = ch != 3 && oL.w + 1u <= oR.w;
            if (a0 || a1 || a2 || a3) {
                if (sp + 4u > S3_UNIT_STACK) { atomicOr(out.flags, S3_FLAG_STACK_OVERFLOW); state = F_DONE; sp = 0; tosValid = false; doSettle = false; }
                else {
                    // depth-first, children in ascending order, the continuation last: the lane goes on in the smallest child that is alive
                    const uint32_t mCont = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)ins << 14) | ((uint32_t)occ << 17);
                    const uint32_t mKid = (uint32_t)s | ((uint32_t)(i + 1) << 3) | ((uint32_t)(ins + 1) << 14) | ((uint32_t)occ << 17);
                    const uint32_t kid = a0 ? 0u : a1 ? 1u : a2 ? 2u : 3u;
                    push(gAl, gAr, gPr, mCont);
                    if (a3 && kid != 3u) push(oL.w + 1u, oR.w, pR, mKid);
                    if (a2 && kid != 2u) push(oL.z + 1u, oR.z, pR - e3, mKid);
                    if (a1 && kid != 1u) push(oL.y + 1u, oR.y, pR - (e2 + e3), mKid);
                    const uint32_t kOc = occAbove(kid, e1, e2, e3);
                    aL = sel4(oL, kid) + 1u; aR = sel4(oR, kid); pR = pR - kOc;
                    ins++; i++;
                    taken = true;
                }
            }
        }
        if (doSettle && !taken) { aL = gAl; aR = gAr; pR = gPr; i++; }
    };

    for (;;) {
        // ---------- retire finished items; the next (case, strand) of the same read, or the read is done ----------
        // (letting finished lanes wait until 8 / 16 / 24 of them can retire and start their next items together was measured in round 5: the first passes of a
        //  ladder 48.9 -> 52.6 / 57.9 / 63.7 ms -- the pass is bound by its waits, not by its instructions)
        if (state == F_DONE) {
            uint32_t sub = subIdx;
            if (goHeavy) {
                // this read goes to the second pass: whatever it wrote so far is ignored there (scatter / check-and-extend test the flag)
                out.heavyFlags[slotIdx] = 1;
                if constexpr (COUNT) nListedRank += readRank;
                if (out.wasHeavy) out.wasHeavy[rIdx] = (uint8_t)in.heavyLevel;  // remembered for the ladder's next phases (see F_LOAD)
                const uint32_t at = atomicAdd(out.heavyCount, 1u);
                out.heavyList[at] = slotIdx;
                out.heavyIndex[slotIdx] = at;
                pendMeta = 0;
                goHeavy = false;
                sub = groups;
            } else if constexpr (TRUNK) {
                out.itemCounts[cntIdx] = seq;
                sub = groups;                                               // one item per turn
            } else {
                out.itemCounts[cntIdx] = seq;
                readTotal += total;
                sub++;
                if (readTotal >= in.maxOut) {
                    // the per-read cut (mergeCountKernel) ends inside this item or before: what follows can never be kept
                    for (; sub < groups; sub++) out.itemCounts[slotIdx * groups + sub] = 0;
                }
            }
            if (sub < groups) {
                const uint32_t cas = sub >> 1;
                subN = sub | ((in.nStepsPacked >> (3u * cas)) & 7u) << 8;
                cntIdx = slotIdx * groups + sub;
                seq = 0; total = 0; sp = 0; tosValid = false;
                state = F_KEYS;
            } else state = F_IDLE;
        }
        // ---------- idle lanes take reads ----------
        unsigned long long idleMask = __ballot(state == F_IDLE);
        if (idleMask && rdyTaken < rdyCount) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
            const uint32_t e = rdyTaken + rank, avail = rdyCount - rdyTaken;
            if (state == F_IDLE && rank < avail) {
                slotIdx = rdy[e]; rIdx = rdy[64u + e];
                subN = ((in.nStepsPacked) & 7u) << 8;
                cntIdx = slotIdx * groups;
                if constexpr (TRUNK) {
                    // queue order: item first, the read last (slot = item * listed reads + read)
                    const uint32_t sub = slotIdx / nListed, cas = sub >> 1;
                    subN = sub | ((in.nStepsPacked >> (3u * cas)) & 7u) << 8;
                    cntIdx = in.unitBase + (listed * groups + sub) * U + (U - 1u);
                }
                readTotal = 0; readCand = 0; readRank = 0; goHeavy = false;
                seq = 0; total = 0; sp = 0; tosValid = false; loadOff = 0;
                state = F_LOAD;
            }
            const uint32_t nIdle = (uint32_t)__popcll(idleMask);
            rdyTaken += nIdle < avail ? nIdle : avail;
            idleMask = __ballot(state == F_IDLE);
        }
        const bool stageA = !queueEmpty && !basePending;
        const bool stageB = basePending && rdyTaken >= rdyCount;
        if (queueEmpty && !basePending && rdyTaken >= rdyCount && idleMask == ~0ULL) break;

        // ---------- lookup-table keys of a starting item (BWTExactModel{Backward,Forward}_Lookup2, BGS-HostAlgnmtAlgo2.cpp:311-347) ----------
        if (state == F_KEYS) {
            s = 0;
            loadStepCache();
            const S3Step st = stepAt(0);
            const int ll = ix.ltSize > st.len ? st.len : ix.ltSize;            // <= 13 characters = 26 bits of one window
            const uint32_t padOnes = ll < ix.ltSize ? ((1u << (2 * (ix.ltSize - ll))) - 1u) : 0u;
            const int shPad = 2 * (ix.ltSize - ll);
            padded = lookupPadded(ix, padFilter, (int)st.type, (int)st.start, (int)st.len, readWindow);
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
            state = F_LT;
        }
        // ---------- what does this lane need?  (addresses only: every lane issues the same loads, back to back) ----------
        uint32_t iL = 0, iR = 0;
        bool needB = false;
        int dir = 0;
        const uint4 *dummy16 = (const uint4 *)ix.pac;
        const uint4 *aP0 = dummy16, *aP1 = dummy16, *aPT = dummy16;
        const uint32_t *aD0 = ix.pac, *aD1 = ix.pac, *aD2 = ix.pac, *aD3 = ix.pac;
        const uint4 *ownLine = ix.fwd.lines;
        if (state == F_STEP) stepAsk(iL, iR, needB, dir, ownLine, aP0, aP1, aD0);
        else if (state == F_LOAD) {
            const uint4 *src = (const uint4 *)(in.readsT + (size_t)rIdx * 2u * wpq + loadOff);   // wpq is a power of two
            aP0 = src;
            if (2u * wpq - loadOff > 4u) aP1 = src + 1;
            if constexpr (!UNIFORM) aD0 = in.readLengths + rIdx;
            if (!TRUNK && out.wasHeavy && loadOff == 0u) aD1 = (const uint32_t *)(out.wasHeavy + (rIdx & ~3u));
        } else if (state == F_LT) {
            const bool bidir = stepAt(0).type != S3_T_BACK_LOOKUP;
            if (ltB == 0xFFFFFFFFu) {
                if (bidir) aP0 = ix.ltBi + ltA;
                else { aD0 = (const uint32_t *)(ix.ltBack + ltA); aD1 = aD0 + 1; }
                if constexpr (COUNT) nLt += 1;
            } else {
                aD0 = ix.lkt + (ltA ? ltA - 1u : 0u);
                aD1 = ix.lkt + ltB;
                if (bidir) { aD2 = ix.rlkt + (ltC ? ltC - 1u : 0u); aD3 = ix.rlkt + ltD; }
                if constexpr (COUNT) nLt += bidir ? 4 : 2;
            }
        }
        const bool running = state == F_STEP || state == F_POPWAIT || state == F_RESUME;
        const bool wantTos = running && !tosValid && sp > 0u;
        if (wantTos) aPT = stk + ((sp - 1u) & (S3_UNIT_STACK - 1u)) * 64u + lane;
        // ---------- issue ----------
        const uint32_t q16 = lane & 3u, q4 = lane >> 2;
        const uint32_t ownLo = (uint32_t)(uintptr_t)ownLine, ownHi = (uint32_t)((uintptr_t)ownLine >> 32);
        auto quadLine = [&](uint32_t owner) -> uint4 {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownLo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)ownHi);
            return loadGlobal16(lo, hi, q16 * 16u);
        };
        const uint4 va0 = quadLine(q4), va1 = quadLine(16u + q4), va2 = quadLine(32u + q4), va3 = quadLine(48u + q4);
        const uint4 p0 = *aP0, p1 = *aP1, pT = *aPT;
        const uint32_t d0 = *aD0, d1 = *aD1, d2 = *aD2, d3 = *aD3;
        // the queue (wave-uniform branches)
        unsigned long long qBase = 0;
        uint32_t cRidx = 0;
        bool cIn = false;
        if (stageA && lane == 0) qBase = atomicAdd(out.workHead, (unsigned long long)S3_UNIT_CHUNK);
        if (stageB) {
            const uint32_t slot = pendBase + lane;
            cIn = slot < nSlots;
            if constexpr (TRUNK) cRidx = cIn ? out.listedRead[slot % nListed] : 0u;
            else cRidx = cIn ? (in.active ? in.active[slot] : slot) : 0u;
        }
        stage[(q4) * 4u + q16] = va0;
        stage[(16u + q4) * 4u + q16] = va1;
        stage[(32u + q4) * 4u + q16] = va2;
        stage[(48u + q4) * 4u + q16] = va3;
        __builtin_amdgcn_wave_barrier();

        if (stageA) {
            const unsigned long long b = (unsigned long long)__shfl((long long)qBase, 0);
            if (b >= nSlots) queueEmpty = true;
            else { pendBase = (uint32_t)b; basePending = true; }
        }
        if (stageB) {
            const uint32_t slot = pendBase + lane;
            if (cIn) { rdy[lane] = slot; rdy[64u + lane] = cRidx; }
            const uint32_t left = nSlots - pendBase;
            rdyCount = left < S3_UNIT_CHUNK ? left : S3_UNIT_CHUNK; rdyTaken = 0;
            basePending = false;
            __builtin_amdgcn_wave_barrier();
        }

        // ---------- advance ----------
        bool doSettle = false, needPop = false;
        if constexpr (COUNT) { nWaveTrips++; nLaneTrips += state != F_IDLE ? 1u : 0u; nStepLane += state == F_STEP ? 1u : 0u; nParked += state == F_RESUME || state == F_POPWAIT ? 1u : 0u; }
        if (wantTos) { tos = pT; tosValid = true; }
        // a read that has cost this many rank queries is heavy whatever it has found so far (dead ends cost steps, not candidates)
        // (a quarter of the budget once the queue is empty -- against the launch's tail -- was measured in round 5: 260 k listed reads instead of 155 k in the
        //  first phase, for the queue is empty while 262 k reads are still under way: 253.5 -> 291.9 ms per step)
        if (!TRUNK && state == F_STEP && readRank > in.heavyRank) { goHeavy = true; state = F_DONE; sp = 0; tosValid = false; }
        if (state == F_STEP) stepTake(iL, iR, needB, dir, p0, p1, d0, doSettle, needPop);
        else if (state == F_RESUME) { doSettle = true; needPop = false; }
        else if (state == F_POPWAIT) { doSettle = true; needPop = true; }
        else if (state == F_LOAD) {
            // A read that was listed at this mismatch budget or a smaller one in an earlier phase of the ladder is listed at once: a read inside a repeat family
            // that found nothing with two mismatches comes back with three, and would spend its rank budget again -- 512 trips of one lane -- before the first
            // pass gives it up (a sixth of the first passes' rank queries, and their tails).  Which reads are listed is nobody's business: the second pass
            // searches them from scratch.
            if (!TRUNK && out.wasHeavy && loadOff == 0u) {
                const uint32_t h = (d1 >> (8u * (rIdx & 3u))) & 0xFFu;
                if (h && h <= in.heavyLevel) { goHeavy = true; state = F_DONE; }
            }
        }
        if (state == F_LOAD) {
            // word gi of the read's row in readsT (both orientations, wpq words each) -> LDS word (orientation * wl + w); words past wl are padding
            const uint32_t vals[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t gi = loadOff + k;
                const uint32_t so = gi >= wpq ? 1u : 0u, w = gi - so * wpq;
                if (gi < 2u * wpq && w < wl) ro[(so * wl + w) * 64u + lane] = vals[k];
            }
            loadOff += 8;
            if constexpr (!UNIFORM) Li = (int)d0;
            if (loadOff >= 2u * wpq) state = F_KEYS;
        } else if (state == F_LT) {
            // the range behind the lookup characters, the way round step 0 moves it
            const bool bidir = stepAt(0).type != S3_T_BACK_LOOKUP;
            int ll = ix.ltSize > sc.len() ? sc.len() : ix.ltSize;
            uint32_t fl, fr, rl, rr;
            if (ltB == 0xFFFFFFFFu) {
                ll = ix.ltK;
                if (!bidir) { fl = d0; fr = d1; rl = fl; rr = fr; }
                else { fl = p0.x; fr = p0.y; rl = p0.z; rr = p0.w; }
            } else if (!bidir) {
                fl = ltA ? d0 + 1u : 1u; fr = d1; rl = fl; rr = fr;
            } else {
                uint32_t l = ltA ? d0 + 1u : 1u, rr_ = d1, rl_ = ltC ? d2 + 1u : 1u, rrr = d3;
                if (rr_ - l < rrr - rl_) rl_ = rrr - (rr_ - l);                          // :349-356
                else if (rr_ - l > rrr - rl_) l = rr_ - (rrr - rl_);
                fl = l; fr = rr_; rl = rl_; rr = rrr;
            }
            if (!bidir) { aL = fl; aR = fr; pR = rr; } else { aL = rl; aR = rr; pR = fr; }
            s = 0; i = ll; ins = 0; occ = 0;
            doSettle = true;
            // (Listing a read at once when one of its lookup strings has thousands of occurrences -- it WILL cost its rank budget, and what it does until then
            //  is thrown away -- was measured in round 5, thresholds 1 Ki .. 64 Ki occurrences: the first passes of a ladder 48.7 -> 38-41 ms, the unit passes
            //  128 -> 159-297 ms.  A read that only brushes a repeat costs the second pass far more than it costs a lane here.)
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
                    if (recCur + rk < recEnd) {
                        out.recs[recCur + rk] = make_uint4(TRUNK ? cntIdx | S3_REC_PASS2 : cntIdx, pendMeta & 0x7FFFFFFFu, pendA, pendB);
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

        // ---------- FAST TRIPS: while at least in.firstFastMin lanes of the wave are stepping, up to in.firstFastTrips trips that do nothing but step; a lane
        // whose next move is anything else parks until the next full trip (s3dp_unit.hip, "FAST TRIPS").  A lane that has spent its read's rank budget
        // stays out: the full trip lists its read ----------
        for (uint32_t ft = 0; ft < in.firstFastTrips; ft++) {
            const bool stepping = state == F_STEP && (TRUNK || readRank <= in.heavyRank);
            if ((uint32_t)__popcll(__ballot(stepping)) < in.firstFastMin) break;
            uint32_t iL = 0, iR = 0;
            bool needB = false;
            int dir = 0;
            const uint4 *ownLine = ix.fwd.lines;
            const uint4 *aP0 = (const uint4 *)ix.pac, *aP1 = (const uint4 *)ix.pac;
            const uint32_t *aD0 = ix.pac;
            if (stepping) stepAsk(iL, iR, needB, dir, ownLine, aP0, aP1, aD0);
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
            if constexpr (COUNT) { nWaveTrips++; nFastWave++; nLaneTrips += state != F_IDLE ? 1u : 0u; nStepLane += stepping ? 1u : 0u; nFastLane += stepping ? 1u : 0u; }
            if (stepping) {
                bool doSettle = false, needPop = false;
                stepTake(iL, iR, needB, dir, p0, p1, d0, doSettle, needPop);
                if (doSettle) {
                    // settle()'s way back to F_STEP, and nothing else of it
                    bool go = false;
                    if (!needPop) {
                        const bool exact = ins >= sc.maxErr();
                        int lenEff = sc.len();
                        if (!exact && sc.minErr() > ins + 1) lenEff -= sc.minErr() - ins - 1;
                        if (i < lenEff && aL <= aR) {
                            const int pos = sc.start() + sc.dir() * i;
                            go = !((aR - aL) < (uint32_t)sc.ceThr() && pos >= sc.ceStart() && pos <= sc.ceEnd());
                            if (in.earlyCe && s == 0 && (aR - aL) < in.earlyCe) go = false;     // (the early hand-over is settle()'s business)
                        }
                    }
                    state = go ? F_STEP : needPop ? F_POPWAIT : F_RESUME;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef listed
#undef sStar
#undef casIdx
#undef subIdx
#undef strand
#undef nSteps
    for (uint32_t k = recCur + lane; k < recEnd; k += 64u) out.recs[k] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if constexpr (COUNT) {
        if (out.counters) {
            atomicAdd(&out.counters[0], (unsigned long long)nRank);
            atomicAdd(&out.counters[1], (unsigned long long)nLt);
            atomicAdd(&out.counters[6], (unsigned long long)nLaneTrips);
            if (lane == 0) atomicAdd(&out.counters[9], (unsigned long long)nWaveTrips);
            atomicAdd(&out.counters[13], (unsigned long long)nStepLane);
            if (lane == 0) atomicAdd(&out.counters[16], (unsigned long long)nFastWave);
            atomicAdd(&out.counters[17], (unsigned long long)nFastLane);
            atomicAdd(&out.counters[18], (unsigned long long)nParked);
            if (lane == 0) { atomicAdd(&out.counters[19], wall_clock64() - tWave0); atomicAdd(&out.counters[20], 1ULL); }
            atomicAdd(&out.counters[21], nListedRank);
        }
    }
}

// trunk = false: first pass, one slot per read of in.nActive; true: trunk pass, one slot per (listed read, item) -- the kernel reads the number of listed
// reads from device memory (in.slotCount); in.nActive then bounds the grid.  Returns a hipError_t.
template <bool TRUNK>
static int launchFirstOrTrunk(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s)
{
    if (in.nActive == 0) return (int)hipSuccess;
    uint64_t blocks = ((uint64_t)in.nActive * (TRUNK ? (uint64_t)in.nCases * 2u : 1u) + S3_SEARCH_BLOCK - 1) / S3_SEARCH_BLOCK;
    if (blocks > maxBlocks) blocks = maxBlocks;
    const size_t lds = S3_V2_MODEL_BYTES + (size_t)(S3_SEARCH_BLOCK / 64) * (4096u + S3_FIRST_RDY_BYTES + 2u * in.wLds * 256u) + v2::S3_PAD_FILTER_BYTES;
    const bool count = out.counters != nullptr;
    const void *fn = in.uniformLen ? (count ? (const void *)firstKernel<true, true, TRUNK> : (const void *)firstKernel<true, false, TRUNK>)
                                   : (count ? (const void *)firstKernel<false, true, TRUNK> : (const void *)firstKernel<false, false, TRUNK>);
    if (lds > 48u * 1024u) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((unsigned)blocks), block(S3_SEARCH_BLOCK);
    if (in.uniformLen) {
        if (count) hipLaunchKernelGGL((firstKernel<true, true, TRUNK>), grid, block, lds, s, ix, in, out);
        else hipLaunchKernelGGL((firstKernel<true, false, TRUNK>), grid, block, lds, s, ix, in, out);
    } else {
        if (count) hipLaunchKernelGGL((firstKernel<false, true, TRUNK>), grid, block, lds, s, ix, in, out);
        else hipLaunchKernelGGL((firstKernel<false, false, TRUNK>), grid, block, lds, s, ix, in, out);
    }
    return (int)hipSuccess;
}

int launchFirstPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s) { return launchFirstOrTrunk<false>(ix, in, out, maxBlocks, s); }
int launchTrunkPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s) { return launchFirstOrTrunk<true>(ix, in, out, maxBlocks, s); }

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadFirst() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::firstKernel<true, false, false>); (void)hipGetLastError(); }
