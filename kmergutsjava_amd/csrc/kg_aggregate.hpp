// kg_aggregate.hpp -- gatherHits (KGJ:457-514) + processSetOfHits (KGJ:385-455) on gfx950.
//
// One wavefront owns one HitContainer (its position-ordered hit records are contiguous in
// hits[]).  The reference's state machine is sequential per container; the wave runs it with
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
// One pass.  A container's CALL records go to its own range of a staging array: a hit votes for at most one CALL (after a
// CALL the list is emptied or cut down to two members that did not vote) and a CALL needs >= minHits voters, so container c
// makes at most hits_c / minHits CALLs and [chs[c] / minHits, chs[c + 1] / minHits) is room enough; the counts are
// prefix-summed and compact_calls_kernel moves the records to calls[] in the reference's emission order, no atomics.
// The pass also marks every record whose vote counted towards a CALL (vote[]): the OTU stage (KGJ:413-439) then is one
// streaming walk over a sequence's records in order -- voters of successive CALLs have ascending indices.
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

__global__ __launch_bounds__(256) void calls_wave_kernel(const kg_hit *__restrict__ hits, const int64_t *__restrict__ chs,
                                                         uint32_t n_cont, AggParams p, uint8_t *acc, uint8_t *vote, uint8_t *tail_ev,
                                                         uint32_t *call_cnt, kg_call *staged /* [n_hits / minHits + 1] */,
                                                         uint32_t per_wave)
{
    // per_wave consecutive containers per wave: 1 for few long containers (contigs); more for millions of short ones
    // (reads), where launching a wave per container would cost more than the work
    const int lane = threadIdx.x & 63;
    const uint32_t c_first = (uint32_t)uni((int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6))) * per_wave;
    // The wave's containers' extents in one coalesced load (per_wave <= 64).  Containers with fewer than two hits need
    // no machine (minHits >= 2: no CALL; a single hit is simply accepted, KGJ:486-497): lanes settle them directly and
    // the sequential part below visits only the others.
    uint32_t my_begin = 0, my_end = 0;
    const bool mine = (uint32_t)lane < per_wave && c_first + (uint32_t)lane < n_cont;
    if (mine) { my_begin = (uint32_t)chs[c_first + lane]; my_end = (uint32_t)chs[c_first + lane + 1]; }
    if (mine && my_end - my_begin < 2) {
        call_cnt[c_first + lane] = 0;
        tail_ev[c_first + lane] = 0;
        if (my_end != my_begin) { acc[my_begin] = (uint8_t)KG_EV_ACCEPTED; vote[my_begin] = 0; }
    }
    uint64_t todo = __ballot(mine && my_end - my_begin >= 2);
    while (todo) {
    const int ci = __builtin_ctzll(todo);
    todo &= todo - 1;
    const uint32_t c = c_first + (uint32_t)ci;
    const uint32_t begin = (uint32_t)rl((int32_t)my_begin, ci), end = (uint32_t)rl((int32_t)my_end, ci);
    kg_call *calls = staged + begin / (uint32_t)p.min_hits;          // the container's staging range (see the header)

    AggState s;
    s.lo = s.last = s.prev = begin;
    s.last_pos = s.last_fI = s.last_avg = s.prev_fI = 0;
    s.cnt = 0; s.currentFI = 0; s.ncalls = 0;
    int32_t carry_pos = 0, carry_fI = 0;            // fields of the record before this chunk
    ChunkRegs pv, ppv;                              // the chunk before the current one and the one before that (registers)
    // "no such chunk": a base no record index comes within 64 of (kg_scan takes < 2^32 - 256 hit records)
    pv.base = ppv.base = 0xFFFFFF00u; pv.mask = ppv.mask = 0; pv.fI = ppv.fI = pv.pos = ppv.pos = 0; pv.wt = ppv.wt = 0.f;

    for (uint32_t base = begin; base < end; base += 64) {
        const int n = (int)min(64u, end - base);
        const uint32_t i = base + lane;
        int32_t pos = 0, fI = 0, avg = 0;
        float wt = 0.f;
        if (lane < n) { pos = hits[i].from0InProt; fI = hits[i].fI; avg = hits[i].avgOffFromEnd; wt = hits[i].functionWt; }
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
    }
    uint32_t tail = 0;
    if (s.cnt >= p.min_hits) {                                                                       // KGJ:511-513
        uint64_t tail_votes = 0;                     // voters in the last chunk, whose bytes are already stored
        tail = process_set(hits, acc, vote, begin, p, s, pv, ppv, tail_votes, c, calls, true) & 1u;
        if ((tail_votes >> lane) & 1ull) vote[pv.base + lane] = 1;
    }
    if (lane == 0) { call_cnt[c] = s.ncalls; tail_ev[c] = (uint8_t)tail; }
    }
}

// staged CALL records -> calls[] in emission order: container c's records sit at staged[chs[c] / minHits ...] and go to
// calls[call_off[c] ...].  W lanes per container (W = 64: few containers with many CALLs; W = 1: millions of reads).
template <int W>
__global__ __launch_bounds__(256) void compact_calls_kernel(const kg_call *__restrict__ staged, const int64_t *__restrict__ chs,
                                                            const uint32_t *__restrict__ call_cnt, const uint32_t *__restrict__ call_off,
                                                            uint32_t n_cont, uint32_t min_hits, kg_call *__restrict__ calls)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t c = t / W;
    if (c >= n_cont) return;
    const uint32_t n = call_cnt[c];
    const kg_call *src = staged + (uint32_t)chs[c] / min_hits;
    kg_call *dst = calls + call_off[c];
    for (uint32_t k = (uint32_t)(t % W); k < n; k += W) dst[k] = src[k];
}

// ccs[c] = call_off[c] widened, plus sentinel
__global__ void call_starts_kernel(const uint32_t *call_off, uint64_t n_cont, const uint64_t *total, int64_t *ccs)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cont) return;
    ccs[c] = c == n_cont ? (int64_t)*total : (int64_t)call_off[c];
}

// OTU vote (KGJ:413-439), one wave per sequence: the records whose vote counted towards a CALL (vote[], left by
// calls_wave_kernel) are replayed in record order -- the reference's order: CALLs of a container are emitted with
// ascending, disjoint voter ranges, containers in order -- against the 5-entry buffer that persists across the
// sequence's containers (KGJ:528, 540).  The buffer lives in wave-uniform registers; the records are streamed in
// 64-record chunks, the next chunk requested before the current one is replayed.
__global__ __launch_bounds__(256) void otu_wave_kernel(const kg_hit *__restrict__ hits, const uint8_t *__restrict__ vote,
                                                       const int64_t *__restrict__ chs, const uint32_t *__restrict__ call_cnt,
                                                       uint32_t n_seqs, uint32_t per, kg_otu *otu, uint32_t per_wave,
                                                       const kg_otu *__restrict__ otu_init)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s_first = (uint32_t)uni((int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6))) * per_wave;
    for (uint32_t s = s_first; s < s_first + per_wave && s < n_seqs; s++) {
    int32_t n = 0;
    int32_t cnt[KG_OI_BUFSZ] = {0, 0, 0, 0, 0}, oi[KG_OI_BUFSZ] = {0, 0, 0, 0, 0};
    if (otu_init) {                                                     // the caller's oICounts (kg_aggregate_hits)
        const kg_otu r0 = otu_init[s];
        n = min(max(r0.n, 0), KG_OI_BUFSZ);
#pragma unroll
        for (int k = 0; k < KG_OI_BUFSZ; k++) { cnt[k] = r0.count[k]; oi[k] = r0.oI[k]; }
    }
    for (uint32_t f = 0; f < per; f++) {
    const uint64_t c = (uint64_t)s * per + f;
    if (call_cnt[c] == 0) continue;                                     // no CALL, no voters (metagenome contigs: almost all)
    const uint32_t begin = (uint32_t)chs[c], end = (uint32_t)chs[c + 1];
    bool n_vote = false;
    int32_t n_o = 0;
    if (begin + (uint32_t)lane < end) { n_vote = vote[begin + lane] != 0; n_o = hits[begin + lane].oI; }
    for (uint32_t b = begin; b < end; b += 64) {
        const bool v = n_vote;
        const int32_t o = n_o;
        n_vote = false; n_o = 0;
        if (b + 64u + (uint32_t)lane < end) { n_vote = vote[b + 64u + lane] != 0; n_o = hits[b + 64u + lane].oI; }
        uint64_t m = __ballot(v);
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            const int32_t ok = rl(o, k);
            int j = n;                                              // KGJ:416-417 linear search
#pragma unroll
            for (int t = KG_OI_BUFSZ - 1; t >= 0; t--)
                if (t < n && oi[t] == ok) j = t;
            if (j == n) {                                           // KGJ:418-427
                if (n == KG_OI_BUFSZ) j--; else n++;
#pragma unroll
                for (int t = 0; t < KG_OI_BUFSZ; t++)
                    if (t == j) { oi[t] = ok; cnt[t] = 1; }
            } else {
#pragma unroll
                for (int t = 0; t < KG_OI_BUFSZ; t++)
                    if (t == j) cnt[t]++;
            }
#pragma unroll
            for (int t = KG_OI_BUFSZ - 1; t >= 1; t--) {            // KGJ:432-437 bubble toward the front
                if (j == t && cnt[t - 1] <= cnt[t]) {
                    int32_t tc = cnt[t - 1], to = oi[t - 1];
                    cnt[t - 1] = cnt[t]; oi[t - 1] = oi[t];
                    cnt[t] = tc; oi[t] = to;
                    j = t - 1;
                }
            }
        }
    }
    }
    if (lane == 0) {
        kg_otu r;
        r.n = n;
#pragma unroll
        for (int k = 0; k < KG_OI_BUFSZ; k++) { r.count[k] = k < n ? cnt[k] : 0; r.oI[k] = k < n ? oi[k] : 0; }
        otu[s] = r;
    }
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
