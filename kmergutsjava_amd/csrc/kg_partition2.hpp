// kg_partition2.hpp -- the second partition level: tags probed out of LDS instead of out of the L2.
//
// Why: bucket_tag_kernel (kg_partition.hpp) probes a bucket's 2 MiB of tags in the XCD's L2, and every 16-byte probe
// moves a whole 128-byte line from the L2 to the CU: 187 GB of line traffic per Gbp for 21.9 GB of tag bytes used; the
// pass sits at the L2's gather ceiling (profiles/r02_pipeline.md).  The way down is fewer lines per probe: the entries
// of a bucket are cut once more, into sub-buckets of 2^sshift slots (32-64 KiB of tags), and a workgroup then loads a
// sub-bucket's tags into LDS ONCE, coalesced, and probes them there.  This is the reference's own plan -- sort the
// queries by slot, then stream the table past them (KGJ:1076-1095, 944-1034) -- taken one level further.
//
//   sub_scatter_kernel (K1): streams the (bucket, scatter workgroup) regions the first level wrote, tile by tile
//                            (<= 4096 entries): one returning LDS atomic per entry gives its rank inside its
//                            sub-bucket, a 64-lane scan + one global atomic per sub-bucket reserve the destination,
//                            the tile is put in sub-bucket order in LDS and leaves in contiguous runs.  Fillers are
//                            dropped.  Nothing is encoded, split or fingerprinted here: 8 bytes in, 8 bytes out.
//   sub_probe_kernel   (K2): one workgroup per sub-bucket at a time: its 2^sshift + 16 tags -> LDS (coalesced
//                            16-byte loads), then per entry one unaligned 16-byte LDS read = the 16-tag window at
//                            the home slot; same decisions and the same candidate records as bucket_tag_kernel,
//                            so verification, overflow handling and ordered placement are shared with the one-level
//                            path, and so are the semantics (KGJ:944-1034: from the home slot forward to the k-mer,
//                            an empty slot or the end of the stream; never wrap).
// A sub-bucket that would outgrow its over-allocated array (heavily repeated k-mers) spills whole 16-entry groups to
// the first level's overflow list, which overflow_probe_kernel probes through the L2 as before.
#pragma once

#include "kg_partition.hpp"

namespace kg {

#ifndef KG_SUB_THREADS
#define KG_SUB_THREADS 512
#endif
#ifndef KG_SUB_SLABS
#define KG_SUB_SLABS 8
#endif
constexpr uint32_t kSubThreads = KG_SUB_THREADS;   // K1 workgroup: 8 waves
constexpr uint32_t kSubSlabs = KG_SUB_SLABS;       // entries per thread and tile
constexpr uint32_t kSubTile = kSubThreads * kSubSlabs;
constexpr uint32_t kMaxSub = 64;            // sub-buckets per bucket (one lane each in the reservation step)

// two consecutive entries with one 16-byte streaming load (p 16-byte aligned)
typedef unsigned long long kg_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_entry_pair(const uint64_t *p, uint64_t &a, uint64_t &b)
{
    const kg_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const kg_u64x2 *>(p));
    a = v.x; b = v.y;
}

// Work item = (bucket b, regions [part * rpi, (part + 1) * rpi) of it); items are drawn from one ticket counter in
// bucket order, so that the workgroups of the grid work on a few buckets at a time and a sub-bucket's array is filled
// by stores that follow each other closely (runs start where the previous one ended: partial lines meet in the L2).
__global__ __launch_bounds__(kSubThreads) void sub_scatter_kernel(
    const uint64_t *__restrict__ ent, const uint32_t *__restrict__ fill, uint32_t n_regions /* per bucket */, uint32_t cap,
    uint32_t n_buckets, uint32_t shift, uint32_t sshift, uint32_t rpi /* regions per item */, uint32_t *next_item /* zeroed */,
    uint64_t *__restrict__ ent2, uint32_t *cur2 /* [n_buckets << (shift - sshift)], zeroed */, uint32_t cap2,
    uint32_t *ovf_cursor, uint32_t ovf_cap, uint32_t *__restrict__ ovf_bucket, uint64_t *__restrict__ ovf_ent)
{
    __shared__ __attribute__((aligned(16))) uint64_t stage[kSubTile];
    __shared__ uint32_t hist[kMaxSub], cnt[kMaxSub], sbase[kMaxSub], gbase[kMaxSub];
    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_sub = 1u << (shift - sshift);
    const uint32_t parts = (n_regions + rpi - 1) / rpi;
    const uint32_t n_items = n_buckets * parts;
    if (tid < kMaxSub) hist[tid] = 0;
    for (;;) {
        __syncthreads();                                   // (also: the previous item's copy-out has read cnt / sbase / gbase)
        if (tid == 0) s_item = atomicAdd(next_item, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= n_items) break;
        const uint32_t b = item / parts, w_lo = (item % parts) * rpi, w_hi = min(w_lo + rpi, n_regions);
        // the item's entries as slabs of kSubThreads consecutive slots of one region; a tile = up to kSubSlabs slabs
        uint32_t w = w_lo, o = 0;                          // next slab: region w, slots [o, o + kSubThreads)   (uniform)
        uint32_t f = w < w_hi ? min(fill[(uint64_t)b * n_regions + w], cap) : 0u;
        uint64_t en[kSubSlabs];
        // (a slab = 2 * kSubThreads consecutive slots, two per thread in one 16-byte load: regions start on 128-byte lines
        //  and hold a multiple of 16 slots)
        static_assert(kSubSlabs % 2 == 0, "entries are loaded in pairs");
        auto load_tile = [&]() {
#pragma unroll
            for (uint32_t k = 0; k < kSubSlabs; k += 2) {
                en[k] = en[k + 1] = kEntInvalid;
                while (w < w_hi && o >= f) {               // next region with entries left (uniform)
                    w++; o = 0;
                    f = w < w_hi ? min(fill[(uint64_t)b * n_regions + w], cap) : 0u;
                }
                if (w < w_hi) {
                    const uint32_t i = o + 2u * tid;
                    if (i < f) {
                        load_entry_pair(ent + ((uint64_t)b * n_regions + w) * cap + i, en[k], en[k + 1]);
                        if (i + 1u >= f) en[k + 1] = kEntInvalid;       // (bulk appends of the low-complexity pass leave odd fills)
                    }
                    o += 2u * kSubThreads;
                }
            }
            while (w < w_hi && o >= f) {                   // (so that the loop ends with the item's last entry, not a tile later)
                w++; o = 0;
                f = w < w_hi ? min(fill[(uint64_t)b * n_regions + w], cap) : 0u;
            }
        };
        bool have = w < w_hi;
        if (have) load_tile();
        while (have) {
            uint64_t e[kSubSlabs];
            uint32_t dr[kSubSlabs];                        // digit << 16 | rank inside the digit (ranks < kSubTile)
#pragma unroll
            for (uint32_t k = 0; k < kSubSlabs; k++) e[k] = en[k];
            have = w < w_hi;
#pragma unroll
            for (uint32_t k = 0; k < kSubSlabs; k++) {
                dr[k] = 0;
                if (e[k] != kEntInvalid) {
                    const uint32_t d = ((uint32_t)e[k] >> sshift) & (n_sub - 1u);
                    dr[k] = (d << 16) | atomicAdd(&hist[d], 1u);
                }
            }
            __syncthreads();
            // one wave: exclusive scan of the tile's histogram (LDS offsets), one global atomic per non-empty sub-bucket; the
            // atomics' round trip to the memory side runs while the tile is put in order in LDS
            uint32_t c_mine = 0, g_mine = 0;
            if (wave == 0) {
                c_mine = lane < n_sub ? hist[lane] : 0u;
                uint32_t incl = c_mine;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t y = __shfl_up(incl, off);
                    if ((int)lane >= off) incl += y;
                }
                if (c_mine) g_mine = atomicAdd(&cur2[(uint64_t)b * n_sub + lane], c_mine);
                cnt[lane] = c_mine; sbase[lane] = incl - c_mine; hist[lane] = 0;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < kSubSlabs; k++)
                if (e[k] != kEntInvalid) stage[sbase[dr[k] >> 16] + (dr[k] & 0xFFFFu)] = e[k];
            if (wave == 0) gbase[lane] = g_mine;
            __syncthreads();
            // runs out: wave v takes the sub-buckets d = v, v + 8, ...; consecutive lanes, consecutive entries
            for (uint32_t d = wave; d < n_sub; d += kSubThreads / 64u) {
                const uint32_t n = cnt[d];
                if (n == 0) continue;
                const uint32_t g = gbase[d], s0 = sbase[d];
                const uint32_t n_fit = g >= cap2 ? 0u : min(n, cap2 - g);
                uint64_t *dst = ent2 + ((uint64_t)b * n_sub + d) * cap2 + g;
                for (uint32_t i = lane; i < n_fit; i += 64u) dst[i] = stage[s0 + i];
                if (n_fit < n) {
                    // the sub-bucket's array is full: whole 16-entry groups to the first level's overflow list
                    const uint32_t rest = n - n_fit, ng = (rest + kGroup - 1) / kGroup;
                    uint32_t g0 = 0;
                    if (lane == 0) g0 = atomicAdd(ovf_cursor, ng);
                    g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0);
                    if (g0 + ng <= ovf_cap) {              // else dropped: the host sees the cursor and falls back
                        for (uint32_t j = lane; j < ng; j += 64u) ovf_bucket[g0 + j] = b;
                        for (uint32_t i = lane; i < ng * kGroup; i += 64u)
                            ovf_ent[(uint64_t)g0 * kGroup + i] = i < rest ? stage[s0 + n_fit + i] : kEntInvalid;
                    }
                }
            }
            // (the next tile's LDS writes to stage[] come after two more barriers)
            // (requesting the next tile before this one is ranked gained nothing alone -- 5.76 against 5.83 ms per Gbp -- and
            //  cost 0.7 ms of the stage beside the scatter pass of the next chunk: profiles/r03_two_level.md)
            if (have) load_tile();
        }
    }
}

// ---------------------------------------------------------------------------------------
// K2.  Item it = (bucket << (shift - sshift)) | sub-bucket: tags [it << sshift, + 2^sshift + 16), entries
// ent2[it * cap2 .. + min(cur2[it], cap2)).  Dynamic LDS: 2^sshift + 16 bytes.
constexpr uint32_t kProbe2Threads = 512;
#ifndef KG_INDEX_THREADS
#define KG_INDEX_THREADS 1024
#endif
constexpr uint32_t kIndexThreads = KG_INDEX_THREADS;      // home-index probe: two workgroups (64 KiB of LDS each) fill a CU's 32 wave slots
static_assert(kProbeN % 2 == 0, "entries are loaded in pairs");

template <bool COUNTERS>
__global__ __launch_bounds__(kProbe2Threads) void sub_probe_kernel(
    const uint8_t *__restrict__ tags, uint64_t limit, const uint64_t *__restrict__ ent2, const uint32_t *__restrict__ cur2,
    uint32_t cap2, uint32_t n_items, uint32_t shift, uint32_t sshift, uint32_t *next_item /* zeroed */,
    CandRec *__restrict__ cand, uint32_t *__restrict__ cand_used, unsigned long long *cand_cursor, uint64_t cand_cap,
    unsigned long long *ctr)
{
    constexpr int N = kProbeN;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u);
    const uint32_t tile_bytes = (1u << sshift) + 16u;
    const uint64_t n_tags = limit + (uint64_t)kTagPad;
    unsigned long long ctr_slots = 0;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;      // "full": the first append takes a chunk
    for (;;) {
        __syncthreads();                                // the previous item's probes have read the tile
        if (tid == 0) s_item = atomicAdd(next_item, 1u);
        __syncthreads();
        const uint32_t it = s_item;
        if (it >= n_items) break;
        const uint32_t n = min(cur2[it], cap2);
        if (n == 0) continue;                           // (uniform)
        const uint64_t t0 = (uint64_t)it << sshift;     // first slot of the sub-bucket
        const uint32_t b = it >> (shift - sshift);
        const uint64_t *src = ent2 + (uint64_t)it * cap2;
        // the first batch of entries is requested in front of the tags (the tags come from HBM or the L2 in whole lines)
        uint64_t ev[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t i = (uint32_t)k * kProbe2Threads + tid;
            ev[k] = i < n ? __builtin_nontemporal_load(src + i) : kEntInvalid;
        }
        for (uint32_t c = tid * 16u; c < tile_bytes; c += kProbe2Threads * 16u) {
            uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);   // behind the tag array: EMPTY
            const uint64_t at = t0 + c;
            if (at + 16u <= n_tags) v = *reinterpret_cast<const uint4 *>(tags + at);
            else if (at < n_tags) {
                uint8_t tmp[16];
                for (uint32_t k = 0; k < 16u; k++) tmp[k] = at + k < n_tags ? tags[at + k] : (uint8_t)kTagEmpty;
                __builtin_memcpy(&v, tmp, 16);
            }
            *reinterpret_cast<uint4 *>(tile + c) = v;
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += kProbe2Threads * N) {
            uint64_t en[N];                             // the next batch, requested before this one is probed
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint32_t i = c0 + kProbe2Threads * N + (uint32_t)k * kProbe2Threads + tid;
                en[k] = i < n ? __builtin_nontemporal_load(src + i) : kEntInvalid;
            }
            uint32_t home[N], id[N], quo[N], fp[N], walked[N];
            bool valid[N];
            Tags16 tg[N];
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint64_t e = ev[k];
                valid[k] = e != kEntInvalid;
                const uint32_t low = (uint32_t)e;
                id[k] = (uint32_t)(e >> 32);
                home[k] = (b << shift) | (low & ((1u << shift) - 1u));       // numSigs < 2^31 on this path
                quo[k] = low >> shift;
                fp[k] = tag_qs(quo[k], home[k]);
                const uint32_t off = low & ((1u << sshift) - 1u);
                if (valid[k]) __builtin_memcpy(&tg[k], tile + off, 16);      // one unaligned ds_read_b128
            }
            uint32_t candm = 0, walkm = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                walked[k] = 0;
                if (valid[k]) {
                    bool emp;
                    const int i = first_stop(tg[k], fp[k], &emp);
                    walked[k] = (uint32_t)i;
                    const uint64_t cur = (uint64_t)home[k] + (uint64_t)i;
                    if (i == 16) { candm |= 1u << k; walkm |= 1u << k; }
                    else if (!emp) candm |= 1u << k;
                    else {
                        if (cur >= limit) ran_off = true;   // the "empty slot" is the padding behind the last record
                        if (COUNTERS) ctr_slots += (cur < limit ? cur + 1 : limit) - (uint64_t)home[k];
                    }
                }
            }
            uint32_t cn[N], rank[N], total = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const unsigned long long m = __ballot((candm >> k) & 1u);
                cn[k] = (uint32_t)__popcll(m);
                rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                total += cn[k];
            }
            if (total) {
                unsigned long long at = chunk_reserve(u, total, cand_used, cand_cursor, cand_cap, lane);
                if (at != ~0ull) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        if ((candm >> k) & 1u) {
                            CandRec cr;
                            cr.home = home[k]; cr.quo = quo[k];
                            cr.id = id[k]; cr.walked = walked[k] | (((walkm >> k) & 1u) ? kWalkOn : 0u);
                            cand[at + rank[k]] = cr;
                        }
                        at += cn[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < N; k++) ev[k] = en[k];
        }
    }
    chunk_finish(u, cand_used, cand_cap, lane);
    flush_ran_off(ran_off, ctr, lane);
    if (COUNTERS) {
        for (int off = 32; off > 0; off >>= 1) ctr_slots += __shfl_down(ctr_slots, off);
        if (lane == 0) atomicAdd(&ctr[1], ctr_slots);
    }
}

// ---------------------------------------------------------------------------------------
// K2 on the home index (kg_device.hpp, build_qidx_kernel): the sub-bucket's 16-bit words -> LDS (2 bytes per slot), then
// per entry ONE aligned 2-byte LDS read and a compare of its quotient with the three listed ones: a match is a hit for
// certain when the index is exact (candidate flagged kScanOn: the verify pass scans the records from the home slot and takes
// the payload; a hashed index sends it through the generic walk instead), no match with the "more" bit clear is a miss for
// certain -- nothing is walked and no fingerprint is computed.  The "more" bit (0.2 %
// of the home slots) falls back to the generic walk (kWalkOn).  lookup_ran_off: a miss whose home slot lies in the
// occupied run that ends at the end of the record stream (home >= tail_start) is a walk that the reference ends with
// an EOFException (KGJ:799-802).  Item it, tile and entries as in sub_probe_kernel; dynamic LDS: 2 << sshift bytes.
__global__ __launch_bounds__(kIndexThreads) void sub_index_kernel(
    const uint16_t *__restrict__ qidx, uint64_t n_idx, uint32_t exact /* every quotient < 31 */, uint32_t tail_start,
    const uint64_t *__restrict__ ent2,
    const uint32_t *__restrict__ cur2, uint32_t cap2, uint32_t n_items, uint32_t shift, uint32_t sshift, uint32_t *next_item /* zeroed */,
    CandRec *__restrict__ cand, uint32_t *__restrict__ cand_used, unsigned long long *cand_cursor, uint64_t cand_cap,
    unsigned long long *ctr)
{
    constexpr int N = kProbeN;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    __shared__ uint32_t s_item;
    const uint16_t *tile16 = reinterpret_cast<const uint16_t *>(tile);
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u);
    const uint32_t tile_slots = 1u << sshift;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;      // "full": the first append takes a chunk
    for (;;) {
        __syncthreads();                                // the previous item's probes have read the tile
        if (tid == 0) s_item = atomicAdd(next_item, 1u);
        __syncthreads();
        const uint32_t it = s_item;
        if (it >= n_items) break;
        const uint32_t n = min(cur2[it], cap2);
        if (n == 0) continue;                           // (uniform)
        const uint64_t t0 = (uint64_t)it << sshift;     // first slot of the sub-bucket
        const uint32_t b = it >> (shift - sshift);
        const uint64_t *src = ent2 + (uint64_t)it * cap2;
        // entries in pairs (one 16-byte load): thread t takes entries 2t, 2t + 1 of every 2 * kIndexThreads (arrays start on
        // 128-byte lines and hold a multiple of 16 slots: the pair behind an odd count is inside the array and ignored)
        auto load_batch = [&](uint32_t c0, uint64_t (&dst)[N]) {
#pragma unroll
            for (int k = 0; k < N; k += 2) {
                const uint32_t i = c0 + (uint32_t)k * kIndexThreads + 2u * tid;
                dst[k] = dst[k + 1] = kEntInvalid;
                if (i < n) {
                    load_entry_pair(src + i, dst[k], dst[k + 1]);
                    if (i + 1u >= n) dst[k + 1] = kEntInvalid;
                }
            }
        };
        uint64_t ev[N];
        load_batch(0u, ev);
        for (uint32_t c = tid * 8u; c < tile_slots; c += kIndexThreads * 8u) {      // eight words = 16 bytes per load
            uint4 v = make_uint4(0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu);  // behind the index: no key
            const uint64_t at = t0 + c;
            if (at + 8u <= n_idx) v = *reinterpret_cast<const uint4 *>(qidx + at);
            *reinterpret_cast<uint4 *>(tile + 2u * c) = v;
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += kIndexThreads * N) {
            uint64_t en[N];                             // the next batch, requested before this one is probed
            load_batch(c0 + kIndexThreads * N, en);
            uint32_t home[N], quo[N], candm = 0, walkm = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint64_t e = ev[k];
                const uint32_t low = (uint32_t)e;
                home[k] = (b << shift) | (low & ((1u << shift) - 1u));       // numSigs < 2^31 on this path
                quo[k] = low >> shift;
                if (e != kEntInvalid) {
                    const uint32_t w = tile16[low & (tile_slots - 1u)];
                    // a 5-bit field of x is zero iff that listed quotient is the query's; (x - 0x0421) & ~x & 0x4210 is non-zero
                    // iff some field is zero (a borrow can only reach a field above a zero one)
                    const uint32_t q5 = exact ? quo[k] : quo[k] % 31u;
                    const uint32_t x = (w & 0x7FFFu) ^ __umul24(q5, 0x0421u);
                    const bool match = (((x - 0x0421u) & ~x) & 0x4210u) != 0u;
                    const bool more = (w & 0x8000u) != 0u;
                    if (match) candm |= 1u << k;
                    else if (more) { candm |= 1u << k; walkm |= 1u << k; }
                    else if (home[k] >= tail_start) ran_off = true;
                }
            }
            uint32_t cn[N], rank[N], total = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const unsigned long long m = __ballot((candm >> k) & 1u);
                cn[k] = (uint32_t)__popcll(m);
                rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                total += cn[k];
            }
            if (total) {
                unsigned long long at = chunk_reserve(u, total, cand_used, cand_cursor, cand_cap, lane);
                if (at != ~0ull) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        if ((candm >> k) & 1u) {
                            CandRec cr;
                            cr.home = home[k]; cr.quo = quo[k];
                            cr.id = (uint32_t)(ev[k] >> 32); cr.walked = (((walkm >> k) & 1u) || !exact) ? kWalkOn : kScanOn;
                            cand[at + rank[k]] = cr;
                        }
                        at += cn[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < N; k++) ev[k] = en[k];
        }
    }
    chunk_finish(u, cand_used, cand_cap, lane);
    flush_ran_off(ran_off, ctr, lane);
}

}  // namespace kg
