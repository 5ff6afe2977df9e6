// kg_aggregate.hpp -- gatherHits (KGJ:457-514) + processSetOfHits (KGJ:385-455) on gfx950.
//
// One wavefront owns one UNIT: a HitContainer (its position-ordered hit records are contiguous in
// hits[]) or, for long containers, a piece of one that starts behind a gap > maxGap (see walk_unit).
// The reference's state machine is sequential per container; the wave runs it with
// wave-uniform (scalar) control flow over 64-record chunks that are loaded coalesced:
//
//   * The reference's "hits" list is always the accepted records inside one index range
//     [lo, last] of the container (it is only ever cleared or cut down to its last two members),
//     so the list is (lo, last, prev, cnt) plus one "accepted" byte per record (the -O order
//     constraint, KGJ:490-494, and the 39 998 cap, KGJ:496, reject records).
//   * FAST path (no -O, list far from the cap): every record is accepted, so the only records at
//     which the machine does more than "append" are those preceded by a gap > maxGap
//     (KGJ:477-484) or carrying the same function index as their predecessor (KGJ:503-508).  Both
//     conditions are per-record facts; one ballot finds them and the scalar loop visits only those.
//   * SLOW path (-O, or the list could reach the cap inside the chunk): record by record.
//   * processSetOfHits walks the list in 64-record chunks; the float32 weight sum is added in list
//     order (KGJ:394) by visiting the voters' lanes in ascending order.
//
// One pass.  A unit's CALL records go to its own range of a staging array: a hit votes for at most one CALL (after a
// CALL the list is emptied or cut down to two members that did not vote) and a CALL needs >= minHits voters, so a unit over
// the records [b, e) makes at most (e - b) / minHits CALLs and [b / minHits, e / minHits) is room enough; the containers'
// counts are prefix-summed and compact_calls_kernel moves the records to calls[] in the reference's emission order.
// The pass also marks every record whose vote counted towards a CALL (vote[]): the OTU stage (KGJ:413-439) pulls those
// records' otuIndex values into a dense list and replays them per sequence -- voters of successive CALLs have ascending
// indices, so record order is the reference's order.
//
// The pass leaves one event byte per record (KG_EV_* in kmerguts_hip.h) and one per
// container: what the machine did at that record (appended it, reset the list before / after it,
// whether that reset printed a CALL and whether it kept the last two members).  Bit 0 is the
// "accepted" byte above; the rest lets the host print the reference's -d stream (HIT / after-hit /
// after-call, KGJ:376-383, 406-409, 470-473, 498-501) without re-deciding anything.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kmerguts_hip.h"

namespace kg {

struct AggParams { int32_t min_hits, min_weighted_hits, max_gap, order_constraint; };

struct AggState {                 // everything here is wave-uniform
    uint32_t lo, last, prev;      // hit indices of the list's first / last / second-to-last member
    int32_t last_pos, last_fI, last_avg, prev_fI;
    int32_t cnt;                  // list size
    int32_t currentFI;
    uint32_t ncalls;
};

__device__ __forceinline__ int32_t rl(int32_t v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// processSetOfHits (KGJ:385-455) on the list [s.lo .. s.last].  chunk grid is anchored at `begin`
// so that the chunk the caller is working on (cur_base, membership bits cur_mask, the lanes' fI and
// functionWt) is taken from registers instead of memory.  A CALL goes to calls[s.ncalls] (the container's staging
// range); its voters are marked in vote[] -- those of the caller's chunk in cur_votes, which the caller stores with the
// chunk's other bytes.
// returns bit 0: a CALL was made, bit 1: the last two members were kept
struct ChunkRegs {               // one 64-record chunk held in registers: base and membership bits wave-uniform, the rest per lane
    uint32_t base;
    uint64_t mask;
    int32_t fI, pos;
    float wt;
};

__device__ __forceinline__ uint32_t process_set(const kg_hit *__restrict__ h, const uint8_t *__restrict__ acc, uint8_t *__restrict__ vote,
                                            uint32_t begin, const AggParams &p, AggState &s, const ChunkRegs &cur, const ChunkRegs &prv,
                                            uint64_t &cur_votes, uint32_t container, kg_call *calls, bool allow_carry)
{
    const int lane = threadIdx.x & 63;
    int32_t fICount = 0;
    float weighted = 0.f;
    uint32_t lastHit = s.lo;
    const uint32_t c0 = begin + ((s.lo - begin) & ~63u);
    for (uint32_t b = c0; b <= s.last; b += 64) {                       // KGJ:390-396
        const uint32_t i = b + lane;
        const bool in = i >= s.lo && i <= s.last;
        int32_t fI = 0;
        float wt = 0.f;
        bool mem = false;
        if (b == cur.base) {                 // the caller's chunk and the one before it are in registers: a set rarely
            fI = cur.fI;                     // reaches further back (runs between gaps: ~40 records in sparse inputs,
            wt = cur.wt;                     // ~13 in dense ones), so most sets are processed without a memory round trip
            mem = in && ((cur.mask >> lane) & 1ull) != 0;
        } else if (b == prv.base) {
            fI = prv.fI;
            wt = prv.wt;
            mem = in && ((prv.mask >> lane) & 1ull) != 0;
        } else if (in) {
            fI = h[i].fI;
            wt = h[i].functionWt;
            mem = (acc[i] & KG_EV_ACCEPTED) != 0;
        }
        uint64_t m = __ballot(in && mem && fI == s.currentFI);
        if (m) {
            fICount += (int32_t)__popcll(m);
            lastHit = b + 63u - (uint32_t)__builtin_clzll(m);
            const int32_t wbits = __float_as_int(wt);
            while (m) {                                                 // float32 sum in list order (KGJ:394)
                const int k = __builtin_ctzll(m);
                m &= m - 1;
                weighted += __int_as_float(rl(wbits, k));
            }
        }
    }
    uint32_t what = 0;
    if (fICount >= p.min_hits && weighted >= (float)p.min_weighted_hits) {      // KGJ:397
        what = 1;
        // positions of the set's first record and of the last voter: out of the register chunks when they are there
        auto pos_of = [&](uint32_t i) -> int32_t {
            if (i - cur.base < 64u) return rl(cur.pos, (int)(i - cur.base));
            if (i - prv.base < 64u) return rl(prv.pos, (int)(i - prv.base));
            return h[i].from0InProt;
        };
        const int32_t pos_lo = pos_of(s.lo), pos_last = pos_of(lastHit);
        if (lane == 0) {
            kg_call c;
            c.container = container;
            c.start = pos_lo;                                           // KGJ:399: first record of the set, any fI
            c.end = pos_last + (KG_K - 1);                              // KGJ:400
            c.count = fICount; c.fI = s.currentFI; c.weightedHits = weighted;
            calls[s.ncalls] = c;
        }
        s.ncalls++;
        // the voters (KGJ:413-415: members with fI == currentFI up to lastHit = all of them)
        for (uint32_t b = c0; b <= s.last; b += 64) {
            const uint32_t i = b + lane;
            const bool in = i >= s.lo && i <= s.last;
            if (b == cur.base) {
                cur_votes |= __ballot(in && ((cur.mask >> lane) & 1ull) != 0 && cur.fI == s.currentFI);
            } else if (b == prv.base) {
                if (in && ((prv.mask >> lane) & 1ull) != 0 && prv.fI == s.currentFI) vote[i] = 1;
            } else if (in && (acc[i] & KG_EV_ACCEPTED) != 0 && h[i].fI == s.currentFI) {
                vote[i] = 1;
            }
        }
    }
    // KGJ:441-453: keep the last two members if they open a new function, else clear
    if (allow_carry && s.cnt >= 2 && s.prev_fI != s.currentFI && s.prev_fI == s.last_fI) {
        s.currentFI = s.last_fI;
        s.lo = s.prev;
        s.cnt = 2;
        what |= 2;
    } else {
        s.cnt = 0;
    }
    return what;
}

// per-chunk event masks (wave-uniform; bit k = record base + k)
struct EvMasks {
    uint64_t pb, pb_call, pb_keep, pa, pa_call, pa_keep;
    __device__ __forceinline__ void before(int k, uint32_t what)
    {
        pb |= 1ull << k;
        pb_call |= (uint64_t)(what & 1) << k;
        pb_keep |= (uint64_t)((what >> 1) & 1) << k;
    }
    __device__ __forceinline__ void after(int k, uint32_t what)
    {
        pa |= 1ull << k;
        pa_call |= (uint64_t)(what & 1) << k;
        pa_keep |= (uint64_t)((what >> 1) & 1) << k;
    }
};

// Long containers are cut into PIECES at static gaps.  One wave per container makes a 4.6 Mbp chromosome (six containers
// of 10^5 hits) a chain of thousands of 64-record chunks: 3-11 ms of aggregation behind a 0.15 ms scan
// (tools/ecoli_time.py).  With the default parameters (no -O, 0 <= maxGap, positions + maxGap below 2^31) a record that lies
// more than maxGap behind its predecessor always finds the machine resetting (KGJ:477-484: the list's last member is at or
// before the predecessor, so the gap rule fires; the list is processed or cleared and the record starts an empty one; no
// carry is possible, see the gap branch below).  What happens behind such a record does not depend on what happened before
// it, so another wave can start there.  piece_starts_kernel picks, per block of 2^pshift records of hits[], at most one such
// record (the first among the block's first 256 records, at least half a block away from both ends of its container); a
// unit = a container's first piece or the piece that starts in block u (calls_wave_kernel).  A unit that runs into the next
// piece's first record does what the gap rule would do there (process or clear its list) and leaves the record's
// RESET_BEFORE / CALL_BEFORE bits in before_ev[u] (merge_before_kernel ORs them into the record's event byte: two waves
// would otherwise write one byte).  CALLs of a piece go to the staging range of its first record; compact_calls_kernel
// walks a container's pieces in order.
constexpr uint32_t kNoPiece = 0xFFFFFFFFu;

// PAIR pieces (round 4; dense inputs have no gaps: sequences assembled from signature k-mers, or a chromosome's coding regions):
// the machine's state is also known, whatever came before, right behind the record p2 at which the pair rule fires (KGJ:503-508):
// the list is processed and cut down to its last two members p1, p2 = two consecutive records of one function Y, currentFI = Y
// (KGJ:441-449) -- exactly the state of a machine that starts with an empty list at p1.  The rule fires at p2 iff currentFI != Y,
// and currentFI is the function of the latest ANCHOR before p1: a record that starts a list (first of its container, or
// behind a gap) or repeats its predecessor's function (after such a record the current function is that record's, whether
// the rule fired there or not).  So a piece may start at p1 when p1 itself is no anchor, p1 + 1 repeats its function without
// a gap, and the latest anchor in the 256 records before p1 carries ANOTHER function.  Only in containers of fewer than
// 39 000 records, where no list comes near the 39 998 cap (KGJ:496: near it the rule looks at records that were not appended).
// The unit that runs into such a piece does at p1 what the pair rule does at p2 -- processSetOfHits on its list, to which p1, p2
// would add no vote (their function is not the current one) -- and hands the record p2 its *_AFTER event bits.
constexpr uint32_t kPairWindow = 256;
constexpr uint32_t kPairMaxContainer = 39000;

__global__ __launch_bounds__(256) void piece_starts_kernel(const kg_hit *__restrict__ hits, const int64_t *__restrict__ chs,
                                                           uint32_t n_hits, uint32_t pshift, int32_t max_gap,
                                                           uint32_t *__restrict__ piece_start, uint8_t *__restrict__ piece_pair /* 1: a pair piece */,
                                                           uint32_t n_blocks, uint32_t pairs_ok)
{
    const int lane = threadIdx.x & 63;
    const uint32_t u = (uint32_t)uni((int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (u >= n_blocks) return;
    uint32_t found = kNoPiece;
    const uint32_t blk = 1u << pshift, i0 = u << pshift;
    if (u > 0 && i0 < n_hits) {
        // a block inside a short container is not worth a piece: look at the container of the block's first record first
        const uint32_t c0 = hits[i0].container;
        const uint32_t lo0 = (uint32_t)chs[c0], hi0 = (uint32_t)chs[c0 + 1];
        const bool long_enough = hi0 - lo0 >= 2u * blk || hi0 < i0 + blk;      // (or the block runs into the next container)
        for (uint32_t j = 0; j < 256u && j < blk && found == kNoPiece && long_enough; j += 64) {
            const uint32_t i = i0 + j + (uint32_t)lane;
            bool gap = false;
            if (i < n_hits && j + (uint32_t)lane < blk) {
                const kg_hit a = hits[i - 1], b = hits[i];
                gap = a.container == b.container && (int32_t)((uint32_t)a.from0InProt + (uint32_t)max_gap) < b.from0InProt;
            }
            const uint64_t m = __ballot(gap);
            if (m) found = i0 + j + (uint32_t)__builtin_ctzll(m);
        }
    }
    if (found != kNoPiece) {
        const uint32_t c = hits[found].container;
        const uint32_t lo = (uint32_t)chs[c], hi = (uint32_t)chs[c + 1];
        if (found - lo < blk / 2 || hi - found < blk / 2) found = kNoPiece;      // pieces of at least half a block
    }
    uint32_t pair = 0;
    if (found == kNoPiece && pairs_ok && u > 0 && i0 < n_hits) {
        const uint32_t c0 = hits[i0].container;
        const uint32_t lo0 = (uint32_t)chs[c0], hi0 = (uint32_t)chs[c0 + 1];
        if (hi0 - lo0 >= 2u * blk && hi0 - lo0 < kPairMaxContainer) {
            const uint32_t w0 = i0 - lo0 > kPairWindow ? i0 - kPairWindow : lo0;          // the look-back starts here
            const uint32_t span = min(256u, blk);                                            // p1 in [i0, i0 + span)
            const uint32_t wend = min(min(i0 + span + 1u, hi0), n_hits);                     // records looked at: [w0, wend)
            int32_t afi = 0;                                                                 // the latest anchor so far (uniform)
            bool have = false, carry_eq = false, carry_start = false;
            for (uint32_t b = w0; b < wend && found == kNoPiece; b += 64) {
                const uint32_t i = b + (uint32_t)lane;
                const bool in = i < wend;
                int32_t fI = 0;
                bool cstart = false, gapf = false, eqf = false;
                if (in) {
                    const kg_hit cur = hits[i];
                    fI = cur.fI;
                    cstart = i == lo0;
                    if (!cstart) {
                        const kg_hit prv = hits[i - 1];
                        gapf = (int32_t)((uint32_t)prv.from0InProt + (uint32_t)max_gap) < cur.from0InProt;
                        eqf = !gapf && prv.fI == cur.fI;
                    }
                }
                const uint64_t am = __ballot(in && (cstart || gapf || eqf)), eqm = __ballot(eqf), stm = __ballot(cstart || gapf);
                // the latest anchor before this lane's record: in this chunk, else the one carried in
                const uint64_t below = am & ((1ull << lane) - 1ull);
                const int al = below ? 63 - __builtin_clzll(below) : 0;
                const int32_t lfi = __shfl(fI, al);
                const bool phave = below ? true : have;
                const int32_t pfi = below ? lfi : afi;
                const bool prev_eq = lane ? ((eqm >> (lane - 1)) & 1ull) != 0 : carry_eq;
                const bool prev_start = lane ? ((stm >> (lane - 1)) & 1ull) != 0 : carry_start;
                // this record is p2 = the second of a run whose first record p1 = i - 1 is no anchor
                const bool cand = eqf && !prev_eq && !prev_start && phave && pfi != fI && i >= i0 + 1u && i - 1u < i0 + span &&
                                  i - 1u - lo0 >= blk / 2 && hi0 - (i - 1u) >= blk / 2;
                const uint64_t cm = __ballot(cand);
                if (cm) { found = b + (uint32_t)__builtin_ctzll(cm) - 1u; pair = 1; }
                if (am) { const int last = 63 - __builtin_clzll(am); afi = rl(fI, last); have = true; }
                carry_eq = (eqm >> 63) & 1ull;
                carry_start = (stm >> 63) & 1ull;
            }
        }
    }
    if (lane == 0) { piece_start[u] = found; piece_pair[u] = (uint8_t)pair; }
}

__global__ void merge_before_kernel(const uint32_t *__restrict__ piece_start, const uint8_t *__restrict__ piece_pair,
                                    const uint8_t *__restrict__ before_ev, uint32_t n_blocks, uint8_t *ev, unsigned long long *n_pieces)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ps = u < n_blocks ? piece_start[u] : kNoPiece;
    // a gap piece: the *_BEFORE bits of its first record; a pair piece: the *_AFTER bits of its second record (where the pair
    // rule fires in the reference)
    if (ps != kNoPiece && before_ev[u]) ev[ps + (piece_pair[u] ? 1u : 0u)] |= before_ev[u];
    const unsigned long long m = __ballot(ps != kNoPiece);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(n_pieces, (unsigned long long)__popcll(m));
}

// One unit: the records [begin, end) of container c, or up to the first record of the next piece.  calls = the unit's
// staging range.  Returns the number of CALLs; *tail_out = the container's tail event when the unit reached `end`.
__device__ __forceinline__ uint32_t walk_unit(const kg_hit *__restrict__ hits, const AggParams &p, uint8_t *acc, uint8_t *vote,
                                              const uint32_t begin, const uint32_t end, const uint32_t c, kg_call *calls,
                                              const uint32_t *__restrict__ piece_start, const uint8_t *__restrict__ piece_pair,
                                              const uint32_t pshift, uint8_t *before_ev, uint32_t *tail_out, bool *reached_end)
{
    const int lane = threadIdx.x & 63;
    uint32_t stopped_at = kNoPiece;
    AggState s;
    s.lo = s.last = s.prev = begin;
    s.last_pos = s.last_fI = s.last_avg = s.prev_fI = 0;
    s.cnt = 0; s.currentFI = 0; s.ncalls = 0;
    int32_t carry_pos = 0, carry_fI = 0;            // fields of the record before this chunk
    ChunkRegs pv, ppv;                              // the chunk before the current one and the one before that (registers)
    // "no such chunk": a base no record index comes within 64 of (kg_scan takes < 2^32 - 256 hit records)
    pv.base = ppv.base = 0xFFFFFF00u; pv.mask = ppv.mask = 0; pv.fI = ppv.fI = pv.pos = ppv.pos = 0; pv.wt = ppv.wt = 0.f;

    // the fields of the next TWO chunks are requested before the current chunk is worked on: a long unit is a chain of
    // chunks, each a dependent step of the machine, and one chunk of look-ahead left every step waiting for most of a
    // memory round trip
    int32_t n_pos = 0, n_fI = 0, n_avg = 0, m_pos = 0, m_fI = 0, m_avg = 0;
    float n_wt = 0.f, m_wt = 0.f;
    if (begin + (uint32_t)lane < end) {
        const kg_hit &h0 = hits[begin + lane];
        n_pos = h0.from0InProt; n_fI = h0.fI; n_avg = h0.avgOffFromEnd; n_wt = h0.functionWt;
    }
    if (begin + 64u + (uint32_t)lane < end) {
        const kg_hit &h1 = hits[begin + 64u + lane];
        m_pos = h1.from0InProt; m_fI = h1.fI; m_avg = h1.avgOffFromEnd; m_wt = h1.functionWt;
    }
    for (uint32_t base = begin; base < end; base += 64) {
        int n = (int)min(64u, end - base);
        if (piece_start) {                           // does another piece start inside this chunk?
            const uint32_t u1 = base >> pshift, u2 = (base + (uint32_t)n - 1u) >> pshift;
            const uint32_t s1 = piece_start[u1];
            if (s1 > begin && s1 >= base && s1 < base + (uint32_t)n) stopped_at = s1;
            if (u2 != u1) {
                const uint32_t s2 = piece_start[u2];
                if (s2 > begin && s2 < base + (uint32_t)n && s2 < stopped_at) stopped_at = s2;
            }
            if (stopped_at != kNoPiece) n = (int)(stopped_at - base);
            if (n == 0) break;
        }
        const uint32_t i = base + lane;
        int32_t pos = n_pos, fI = n_fI, avg = n_avg;
        float wt = n_wt;
        if (lane >= n) { pos = 0; fI = 0; avg = 0; wt = 0.f; }      // (a chunk cut short by the next piece)
        n_pos = m_pos; n_fI = m_fI; n_avg = m_avg; n_wt = m_wt;
        m_pos = 0; m_fI = 0; m_avg = 0; m_wt = 0.f;
        if (base + 128u + (uint32_t)lane < end) {
            const kg_hit &h2 = hits[base + 128u + lane];
            m_pos = h2.from0InProt; m_fI = h2.fI; m_avg = h2.avgOffFromEnd; m_wt = h2.functionWt;
        }
        uint64_t accmask;
        uint64_t votes = 0;                          // records of this chunk whose vote counted towards a CALL
        EvMasks em = {0, 0, 0, 0, 0, 0};
        ChunkRegs cu;                                // (cu.mask follows accmask at every call)
        cu.base = base; cu.mask = 0; cu.fI = fI; cu.pos = pos; cu.wt = wt;

        // the fast path needs every record of the chunk to be accepted and the list's last member to be
        // the record just before the chunk (after a cap overflow the list can end far behind)
        const bool fast = !p.order_constraint && s.cnt + n < KG_MAX_HITS_PER_SEQ - 2 &&
                          (s.cnt == 0 || s.last + 1 == base);
        if (fast) {
            accmask = n == 64 ? ~0ull : ((1ull << n) - 1ull);
            int32_t ppos = __shfl_up(pos, 1), pfI = __shfl_up(fI, 1);
            if (lane == 0) { ppos = carry_pos; pfI = carry_fI; }
            const bool first = i == begin;
            // KGJ:477-478 with Java int wrap-around: last.from0InProt + maxGap < ph.from0InProt
            const bool gapf = !first && (int32_t)((uint32_t)ppos + (uint32_t)p.max_gap) < pos;
            const bool eqf = !first && fI == pfI;
            const uint64_t gapm = __ballot(gapf), eqm = __ballot(eqf);
            const uint64_t inm = n == 64 ? ~0ull : ((1ull << n) - 1ull);
            int k0 = 0;
            for (;;) {
                // The next record at which the machine does more than append: one behind a gap, or one that repeats its
                // predecessor's function while that function is not the current one (KGJ:503-508).  The current
                // function only changes at such records (or when an empty list restarts at k0), so in dense inputs --
                // long runs of the current function -- whole runs are skipped with one ballot.
                if (k0 >= n) break;
                const int32_t cfi = s.cnt > 0 ? s.currentFI : rl(fI, k0);
                const uint64_t neqm = __ballot(fI != cfi);
                const uint64_t ev = (gapm | (eqm & neqm)) & inm & ~((1ull << k0) - 1ull);
                if (!ev) break;
                const int k = __builtin_ctzll(ev);
                if (k > k0) {                                           // records k0..k-1: plain appends
                    if (s.cnt == 0) { s.currentFI = rl(fI, k0); s.lo = base + k0; }     // KGJ:486-488
                    s.cnt += k - k0;
                }
                const int32_t fk = rl(fI, k);
                const uint32_t ik = base + (uint32_t)k;
                if (s.cnt > 0 && ((gapm >> k) & 1)) {                                   // KGJ:477-484
                    uint32_t what = 0;
                    if (s.cnt >= p.min_hits) {
                        s.last = ik - 1;
                        // no carry is possible here: a pair of equal, non-current fI at the end of the list
                        // would have fired the pair rule when its second record was appended
                        what = (cu.mask = accmask, process_set(hits, acc, vote, begin, p, s, cu, pv, votes, c, calls, false));
                    } else {
                        s.cnt = 0;
                    }
                    em.before(k, what);
                }
                if (s.cnt == 0) { s.currentFI = fk; s.lo = ik; }                         // KGJ:486-488
                s.cnt++;                                                                 // KGJ:496-497
                if (s.cnt > 1 && s.currentFI != fk && ((eqm >> k) & 1)) {                // KGJ:503-508
                    s.last = ik; s.prev = ik - 1; s.last_fI = fk; s.prev_fI = fk;
                    em.after(k, (cu.mask = accmask, process_set(hits, acc, vote, begin, p, s, cu, pv, votes, c, calls, true)));
                }
                k0 = k + 1;
            }
            if (n > k0) {
                if (s.cnt == 0) { s.currentFI = rl(fI, k0); s.lo = base + k0; }
                s.cnt += n - k0;
            }
            // hand-over state for a following chunk (which may take the slow path)
            if (s.cnt > 0) {
                s.last = base + n - 1;
                s.last_pos = rl(pos, n - 1); s.last_fI = rl(fI, n - 1); s.last_avg = rl(avg, n - 1);
                if (s.cnt > 1) { s.prev = s.last - 1; s.prev_fI = n >= 2 ? rl(fI, n - 2) : carry_fI; }
            }
        } else {
            accmask = 0;
            for (int k = 0; k < n; k++) {
                const int32_t pk = rl(pos, k), fk = rl(fI, k), ak = rl(avg, k);
                const uint32_t ik = base + (uint32_t)k;
                if (s.cnt > 0 && (int32_t)((uint32_t)s.last_pos + (uint32_t)p.max_gap) < pk) {      // KGJ:477-484
                    uint32_t what = 0;
                    if (s.cnt >= p.min_hits)
                        what = (cu.mask = accmask, process_set(hits, acc, vote, begin, p, s, cu, pv, votes, c, calls, true));
                    else
                        s.cnt = 0;
                    em.before(k, what);
                }
                if (s.cnt == 0) s.currentFI = fk;                                                    // KGJ:486-488
                bool ok = !p.order_constraint || s.cnt == 0;
                if (!ok) {                                                                           // KGJ:490-494
                    const int32_t d = (int32_t)((uint32_t)(pk - s.last_pos) - (uint32_t)(s.last_avg - ak));
                    const int32_t ad = d < 0 ? (int32_t)(0u - (uint32_t)d) : d;                     // Math.abs(int)
                    ok = fk == s.last_fI && ad <= 20;
                }
                if (ok) {
                    if (s.cnt < KG_MAX_HITS_PER_SEQ - 2) {                                           // KGJ:496-497
                        if (s.cnt == 0) { s.lo = ik; s.prev = ik; s.prev_fI = fk; }
                        else { s.prev = s.last; s.prev_fI = s.last_fI; }
                        s.last = ik; s.last_pos = pk; s.last_fI = fk; s.last_avg = ak;
                        s.cnt++;
                        accmask |= 1ull << k;
                    }
                    if (s.cnt > 1 && s.currentFI != fk && s.prev_fI == s.last_fI)                    // KGJ:503-508
                        em.after(k, (cu.mask = accmask, process_set(hits, acc, vote, begin, p, s, cu, pv, votes, c, calls, true)));
                }
            }
        }
        if (lane < n) {
            uint32_t e = (uint32_t)((accmask >> lane) & 1ull) * KG_EV_ACCEPTED;
            e |= (uint32_t)((em.pb >> lane) & 1ull) * KG_EV_RESET_BEFORE;
            e |= (uint32_t)((em.pb_call >> lane) & 1ull) * KG_EV_CALL_BEFORE;
            e |= (uint32_t)((em.pb_keep >> lane) & 1ull) * KG_EV_KEEP2_BEFORE;
            e |= (uint32_t)((em.pa >> lane) & 1ull) * KG_EV_RESET_AFTER;
            e |= (uint32_t)((em.pa_call >> lane) & 1ull) * KG_EV_CALL_AFTER;
            e |= (uint32_t)((em.pa_keep >> lane) & 1ull) * KG_EV_KEEP2_AFTER;
            acc[i] = (uint8_t)e;
            vote[i] = (uint8_t)((votes >> lane) & 1ull);
        }
        carry_pos = rl(pos, n - 1);
        carry_fI = rl(fI, n - 1);
        ppv = pv;
        cu.mask = accmask;
        pv = cu;
        if (stopped_at != kNoPiece) break;
    }
    uint32_t tail = 0;
    if (stopped_at != kNoPiece && piece_pair[stopped_at >> pshift]) {
        // the next piece starts at the first record of a pair whose second record fires the pair rule (KGJ:503-508; see
        // piece_starts_kernel): processSetOfHits on the list as it stands -- the pair's two records would be its last members and
        // cast no vote -- after which the reference keeps exactly those two: the next piece's business.  The event bits go
        // to the pair's second record.
        uint32_t bits = 0;
        if (s.cnt > 0) {
            bits = KG_EV_RESET_AFTER | KG_EV_KEEP2_AFTER;
            uint64_t tail_votes = 0;
            if (process_set(hits, acc, vote, begin, p, s, pv, ppv, tail_votes, c, calls, false) & 1u) bits |= KG_EV_CALL_AFTER;
            if ((tail_votes >> lane) & 1ull) vote[pv.base + lane] = 1;
        }
        if (lane == 0) before_ev[stopped_at >> pshift] = (uint8_t)bits;
    } else if (stopped_at != kNoPiece) {
        // the next piece's first record lies behind a gap: what the gap rule does there (KGJ:477-484), no carry
        uint32_t bits = 0;
        if (s.cnt > 0) {
            bits = KG_EV_RESET_BEFORE;
            if (s.cnt >= p.min_hits) {
                uint64_t tail_votes = 0;
                if (process_set(hits, acc, vote, begin, p, s, pv, ppv, tail_votes, c, calls, false) & 1u) bits |= KG_EV_CALL_BEFORE;
                if ((tail_votes >> lane) & 1ull) vote[pv.base + lane] = 1;
            }
        }
        if (lane == 0) before_ev[stopped_at >> pshift] = (uint8_t)bits;
    } else if (s.cnt >= p.min_hits) {                                                                // KGJ:511-513
        uint64_t tail_votes = 0;                     // voters in the last chunk, whose bytes are already stored
        tail = process_set(hits, acc, vote, begin, p, s, pv, ppv, tail_votes, c, calls, true) & 1u;
        if ((tail_votes >> lane) & 1ull) vote[pv.base + lane] = 1;
    }
    *tail_out = tail;
    *reached_end = stopped_at == kNoPiece;
    return s.ncalls;
}

// One wave per unit.  Waves [0, n_cwaves): first pieces, per_wave consecutive containers per wave (1 for contigs; more
// when there are millions of short containers -- reads -- where a wave per container would cost more than the work).
// Waves behind them: the piece that starts in block u = wave - n_cwaves, if any.  call_cnt[] (the containers' CALL
// totals) is zeroed by the caller; first_cnt[c] / piece_cnt[u] = the CALLs of a container's first piece / of block u's.
__global__ __launch_bounds__(256) void calls_wave_kernel(const kg_hit *__restrict__ hits, const int64_t *__restrict__ chs,
                                                         uint32_t n_cont, AggParams p, uint8_t *acc, uint8_t *vote, uint8_t *tail_ev,
                                                         uint32_t *call_cnt, uint32_t *first_cnt, kg_call *staged /* [n_hits / minHits + 1] */,
                                                         uint32_t per_wave, uint32_t n_cwaves, const uint32_t *__restrict__ piece_start,
                                                         uint32_t pshift, uint32_t n_pblocks, uint32_t *piece_cnt, uint8_t *before_ev,
                                                         const uint8_t *__restrict__ piece_pair)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)uni((int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (w >= n_cwaves) {
        const uint32_t u = w - n_cwaves;
        if (u >= n_pblocks) return;
        const uint32_t begin = piece_start[u];
        if (begin == kNoPiece) return;
        const uint32_t c = hits[begin].container;
        const uint32_t end = (uint32_t)chs[c + 1];
        uint32_t tail;
        bool reached_end;
        const uint32_t ncalls = walk_unit(hits, p, acc, vote, begin, end, c, staged + begin / (uint32_t)p.min_hits, piece_start, piece_pair,
                                          pshift, before_ev, &tail, &reached_end);
        if (lane == 0) {
            piece_cnt[u] = ncalls;
            if (ncalls) atomicAdd(&call_cnt[c], ncalls);
            if (reached_end) tail_ev[c] = (uint8_t)tail;
        }
        return;
    }
    const uint32_t c_first = w * per_wave;
    // The wave's containers' extents in one coalesced load (per_wave <= 64).  Containers with fewer than two hits need
    // no machine (minHits >= 2: no CALL; a single hit is simply accepted, KGJ:486-497): lanes settle them directly and
    // the sequential part below visits only the others.
    uint32_t my_begin = 0, my_end = 0;
    const bool mine = (uint32_t)lane < per_wave && c_first + (uint32_t)lane < n_cont;
    if (mine) { my_begin = (uint32_t)chs[c_first + lane]; my_end = (uint32_t)chs[c_first + lane + 1]; }
    if (mine && my_end - my_begin < 2) {
        first_cnt[c_first + lane] = 0;
        tail_ev[c_first + lane] = 0;
        if (my_end != my_begin) { acc[my_begin] = (uint8_t)KG_EV_ACCEPTED; vote[my_begin] = 0; }
    }
    uint64_t todo = __ballot(mine && my_end - my_begin >= 2);
    while (todo) {
        const int ci = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t c = c_first + (uint32_t)ci;
        const uint32_t begin = (uint32_t)rl((int32_t)my_begin, ci), end = (uint32_t)rl((int32_t)my_end, ci);
        uint32_t tail;
        bool reached_end;
        const uint32_t ncalls = walk_unit(hits, p, acc, vote, begin, end, c, staged + begin / (uint32_t)p.min_hits, piece_start, piece_pair,
                                          pshift, before_ev, &tail, &reached_end);
        if (lane == 0) {
            first_cnt[c] = ncalls;
            if (ncalls) atomicAdd(&call_cnt[c], ncalls);
            if (reached_end) tail_ev[c] = (uint8_t)tail;
        }
    }
}

// staged CALL records -> calls[] in emission order: the records of container c's first piece sit at
// staged[chs[c] / minHits ...], those of a later piece at staged[its first record / minHits ...]; all go to
// calls[call_off[c] ...], piece after piece.  W lanes per container (W = 64: few containers with many CALLs; W = 1: millions
// of reads).
template <int W>
__global__ __launch_bounds__(256) void compact_calls_kernel(const kg_call *__restrict__ staged, const int64_t *__restrict__ chs,
                                                            const uint32_t *__restrict__ first_cnt, const uint32_t *__restrict__ call_off,
                                                            uint32_t n_cont, uint32_t min_hits, kg_call *__restrict__ calls,
                                                            const uint32_t *__restrict__ piece_start, const uint32_t *__restrict__ piece_cnt,
                                                            uint32_t pshift)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t c = t / W;
    if (c >= n_cont) return;
    const uint32_t begin = (uint32_t)chs[c], end = (uint32_t)chs[c + 1];
    kg_call *dst = calls + call_off[c];
    {
        const uint32_t n = first_cnt[c];
        const kg_call *src = staged + begin / min_hits;
        for (uint32_t k = (uint32_t)(t % W); k < n; k += W) dst[k] = src[k];
        dst += n;
    }
    if (piece_start && end > begin) {
        if (W == 64) {
            // a chromosome's container has hundreds of pieces with a few CALLs each: a lane per piece, their places by a wave scan
            const int lane = threadIdx.x & 63;
            const uint32_t last = (end - 1u) >> pshift;
            for (uint32_t u0 = begin >> pshift; u0 <= last; u0 += 64) {
                const uint32_t u = u0 + (uint32_t)lane;
                const uint32_t ps = u <= last ? piece_start[u] : kNoPiece;
                const bool mine = ps > begin && ps < end;
                const uint32_t n = mine ? piece_cnt[u] : 0u;
                uint32_t incl = n;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
                    if (lane >= off) incl += v;
                }
                const kg_call *src = staged + (mine ? ps / min_hits : 0u);
                for (uint32_t k = 0; k < n; k++) dst[incl - n + k] = src[k];
                dst += (uint32_t)__shfl((int)incl, 63);
            }
        } else {
            for (uint32_t u = begin >> pshift; u <= (end - 1u) >> pshift; u++) {
                const uint32_t ps = piece_start[u];
                if (ps <= begin || ps >= end) continue;
                const uint32_t n = piece_cnt[u];
                const kg_call *src = staged + ps / min_hits;
                for (uint32_t k = (uint32_t)(t % W); k < n; k += W) dst[k] = src[k];
                dst += n;
            }
        }
    }
}

// ccs[c] = call_off[c] widened, plus sentinel
__global__ void call_starts_kernel(const uint32_t *call_off, uint64_t n_cont, const uint64_t *total, int64_t *ccs)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cont) return;
    ccs[c] = c == n_cont ? (int64_t)*total : (int64_t)call_off[c];
}

// OTU vote (KGJ:413-439).  The records whose vote counted towards a CALL (vote[], left by calls_wave_kernel) are replayed
// in record order -- the reference's order: CALLs of a container are emitted with ascending, disjoint voter ranges,
// containers in order -- against the 5-entry buffer that persists across the sequence's containers (KGJ:528, 540).  The
// replay is sequential per sequence, but only over the voters: two small parallel kernels pull the voters' otuIndex values
// out of hits[] into one dense list first (count per 64-record chunk, prefix sum, scatter); a wave that walked all of a
// chromosome's 10^5-10^6 records to find them took 2-6 ms (tools/ecoli_time.py).
__global__ __launch_bounds__(256) void voter_count_kernel(const uint8_t *__restrict__ vote, uint32_t n_hits, uint32_t *__restrict__ cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t m = __ballot(i < n_hits && vote[i] != 0);
    if ((threadIdx.x & 63) == 0 && (i & ~63u) < n_hits) cnt[i >> 6] = (uint32_t)__popcll(m);
    if (i == 0) cnt[(n_hits + 63u) >> 6] = 0;              // the item behind the last chunk (its prefix = the total)
}

__global__ __launch_bounds__(256) void voter_scatter_kernel(const kg_hit *__restrict__ hits, const uint8_t *__restrict__ vote,
                                                            uint32_t n_hits, const uint32_t *__restrict__ voff, int32_t *__restrict__ vlist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool v = i < n_hits && vote[i] != 0;
    const uint64_t m = __ballot(v);
    if (v) vlist[voff[i >> 6] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = hits[i].oI;
}

// one wave per sequence (per_wave consecutive sequences per wave when there are millions of short ones)
__global__ __launch_bounds__(256) void otu_wave_kernel(const int32_t *__restrict__ vlist, const uint32_t *__restrict__ voff,
                                                       const uint8_t *__restrict__ vote, const int64_t *__restrict__ chs,
                                                       uint32_t n_hits, uint32_t n_seqs, uint32_t per, kg_otu *otu, uint32_t per_wave,
                                                       const kg_otu *__restrict__ otu_init)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s_first = (uint32_t)uni((int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6))) * per_wave;
    // index in vlist[] of the first voter at or behind record i
    auto voter_index = [&](uint32_t i) -> uint32_t {
        const uint32_t c0 = i & ~63u;
        const uint64_t m = __ballot(c0 + (uint32_t)lane < n_hits && c0 + (uint32_t)lane < i && vote[c0 + lane] != 0);
        return (c0 < n_hits ? voff[c0 >> 6] : voff[(n_hits + 63u) >> 6]) + (uint32_t)__popcll(m);
    };
    // The five-entry buffer lives in the LANES 0..4 of the wave (round 4; in scalar registers before: ~100 dependent scalar
    // instructions per vote, 0.83 of the 1.43 ms of aggregation on the high-density config and most of a single chromosome's):
    // lane t holds entry t; a vote is one ballot (where is it?), one ballot (how far does it bubble?) and a rotate by DPP.
    for (uint32_t s = s_first; s < s_first + per_wave && s < n_seqs; s++) {
        int32_t n = 0;                                                      // (wave-uniform)
        uint32_t nmask = 0;                                                 // (1 << n) - 1: the lanes that hold an entry
        int32_t cnt = 0, oi = 0;                                            // lane t < n: entry t
        if (otu_init) {                                                     // the caller's oICounts (kg_aggregate_hits)
            const kg_otu *r0 = otu_init + s;
            n = uni(min(max(r0->n, 0), KG_OI_BUFSZ));
            nmask = (1u << n) - 1u;
            if (lane < KG_OI_BUFSZ) { cnt = r0->count[lane]; oi = r0->oI[lane]; }
        }
        const uint32_t vb = voter_index((uint32_t)chs[(uint64_t)s * per]), ve = voter_index((uint32_t)chs[(uint64_t)(s + 1) * per]);
        int32_t n_o = vb + (uint32_t)lane < ve ? vlist[vb + lane] : 0;
        for (uint32_t b = vb; b < ve; b += 64) {
            const int32_t o = n_o;
            n_o = b + 64u + (uint32_t)lane < ve ? vlist[b + 64u + lane] : 0;     // the next chunk, before this one is replayed
            const int nv = (int)min(64u, ve - b);
            // Consecutive voters with the same otuIndex (a genome's hits mostly name its own OTU) are replayed as one step
            // of their count: r single steps = add r, then one pass of the bubble (an entry only ever moves forward, past the
            // neighbours whose count does not exceed its own, and its count only grows -- the same neighbours either way;
            // a new entry likewise: inserted with 1 and raised r - 1 times = inserted with r).
            const int32_t before = __shfl_up(o, 1);
            uint64_t heads = __ballot(lane < nv && (lane == 0 || o != before));
            // a head lane's run length, computed by all lanes at once (one readlane per step instead of five scalar instructions)
            const uint64_t above = lane < 63 ? heads >> (lane + 1) : 0ull;
            const int32_t rlen = above ? __builtin_ctzll(above) + 1 : nv - lane;
            // With one wave per SIMD every instruction of a step is on the clock (config 5: 3 300 steps per sequence, nothing
            // else to run): ballots of ONE compare each, masked with scalar masks (a ballot of "lane < n && ..." costs two more
            // VALU instructions); and the common outcome of the bubble -- the entry in front has more, nothing moves -- is one
            // bit test.
            while (heads) {
                const int k = __builtin_ctzll(heads);
                heads &= heads - 1;
                const int32_t ok = rl(o, k), r = rl(rlen, k);
                const uint64_t at = __ballot(oi == ok) & nmask;             // KGJ:416-417 linear search: the FIRST entry that
                int j;                                                      // carries it (a caller's buffer may hold it twice)
                if (at) {
                    j = __builtin_ctzll(at);
                    cnt += lane == j ? r : 0;
                } else {                                                    // KGJ:418-427: append, or overwrite the last entry
                    if (n == KG_OI_BUFSZ) j = KG_OI_BUFSZ - 1; else { j = n++; nmask = nmask * 2u + 1u; }
                    if (lane == j) { oi = ok; cnt = r; }
                }
                if (j == 0) continue;                                       // the leading entry: nothing to bubble past
                // KGJ:432-437: toward the front while the neighbour in front does not have MORE: past the run of entries with
                // cnt <= mine that ends right in front of j (for a buffer this replay built itself the counts never increase
                // towards the back, but a caller's oICounts may hold anything)
                const int32_t cj = rl(cnt, j);
                const uint32_t more = (uint32_t)__ballot(cnt > cj);          // (bits at and behind j: not looked at)
                if ((more >> (j - 1)) & 1u) continue;                        // the neighbour in front has more: entry j stays
                const uint32_t gt = more & ((1u << j) - 1u);
                const int p = gt ? 32 - __builtin_clz(gt) : 0;               // new place of entry j (< j)
                const int32_t oj = rl(oi, j);
                // the neighbour in front through DPP (row_shr:1; the five lanes share a row): no LDS round trip
                const int32_t pc = __builtin_amdgcn_update_dpp(0, cnt, 0x111, 0xF, 0xF, false);
                const int32_t po = __builtin_amdgcn_update_dpp(0, oi, 0x111, 0xF, 0xF, false);
                if (lane > p && lane <= j) { cnt = pc; oi = po; }
                else if (lane == p) { cnt = cj; oi = oj; }
            }
        }
        // lane t writes entry t (zeros behind n), lane 0 the count as well
        if (lane < KG_OI_BUFSZ) {
            otu[s].count[lane] = lane < n ? cnt : 0;
            otu[s].oI[lane] = lane < n ? oi : 0;
        }
        if (lane == 0) otu[s].n = n;
    }
}

// processSetOfHits (KGJ:385-455) on one caller-supplied list, as the public method of the reference class does it:
// one lane walks the list (the method is a single step of gatherHits' state machine; kg_process_set_of_hits).
// out[0] = 1 if a CALL was made, out[1] = the returned currentFI, out[2] = 1 if the list keeps its last two members.
__global__ void process_set_single_kernel(const kg_hit *__restrict__ hits, int32_t n, int32_t current_fi, AggParams p,
                                          kg_otu *otu, kg_call *call, int32_t *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t count = 0, last = 0;
    float wt = 0.f;
    for (int32_t i = 0; i < n; i++)                                        // KGJ:390-396
        if (hits[i].fI == current_fi) { last = i; count++; wt += hits[i].functionWt; }
    int32_t called = 0;
    kg_otu o = *otu;
    if (count >= p.min_hits && wt >= (float)p.min_weighted_hits) {         // KGJ:397
        called = 1;
        kg_call c;
        c.container = hits[0].container; c.start = hits[0].from0InProt; c.end = hits[last].from0InProt + (KG_K - 1);
        c.count = count; c.fI = current_fi; c.weightedHits = wt;
        *call = c;
        for (int32_t i = 0; i <= last; i++) {                              // KGJ:413-439
            if (hits[i].fI != current_fi) continue;
            int32_t j = 0;
            while (j < o.n && o.oI[j] != hits[i].oI) j++;
            if (j == o.n) {
                if (o.n == KG_OI_BUFSZ) j--; else o.n++;
                o.oI[j] = hits[i].oI; o.count[j] = 1;
            } else o.count[j]++;
            while (j > 0 && o.count[j - 1] <= o.count[j]) {
                const int32_t tc = o.count[j - 1], to = o.oI[j - 1];
                o.count[j - 1] = o.count[j]; o.oI[j - 1] = o.oI[j];
                o.count[j] = tc; o.oI[j] = to;
                j--;
            }
        }
    }
    *otu = o;
    const bool keep2 = hits[n - 2].fI != current_fi && hits[n - 2].fI == hits[n - 1].fI;      // KGJ:441-449
    out[0] = called; out[1] = keep2 ? hits[n - 1].fI : current_fi; out[2] = keep2 ? 1 : 0;
}

}  // namespace kg
