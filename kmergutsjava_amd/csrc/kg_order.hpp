// kg_order.hpp -- ordered placement of the partitioned scan's hits without a random access.
//
// The partitioned scan finds its hits in table-slot order, i.e. in random order with respect to the windows they
// belong to; hits[] has to be ordered by (container, from0InProt) (the reference sorts each container's hits,
// KGJ:460-465).  Until round 2 every hit set a bit in a 64-bit mask of its (block, row) with a global atomicOr, read a
// 24-byte row record and was stored at off[row] + popcount(mask bits before it): three random 128-byte lines per
// 24-byte record (21.7 GB of traffic to deliver 0.88 GB of hit records per Gbp), 5.2 ms of a 20.7 ms stage.
//
// Now every query entry carries, instead of (block, row, lane), its KEY = the window's rank in the final order:
//     key = row_index * 64 + olane,   row_index = index of the window's 64-window row in the container-major row order
//                                     (kg_device.hpp, row_index: rows of one container are consecutive and ascend with
//                                     the position), olane = the window's rank inside its row by position
//                                     ('+' rows and proteins: the lane; '-' rows: 63 - lane)
// so that ascending key IS (container, from0InProt) order, and the unordered hit list {key, payload} is ordered by a
// two-level partition by key range followed by an in-LDS ranking, all of it streaming:
//   hit_hist_kernel        hits per GROUP (2^gshift rows = 2^(gshift+6) keys; <= kMaxGroups groups per chunk of the batch),
//                          accumulated in LDS by persistent workgroups, one global atomic per (workgroup, non-empty group)
//   group_scan_kernel      exclusive scan of the group counts: where every group starts -- exact, nothing is over-allocated
//   hit_partition_kernel   <first> by the high part of the group number (groups / 128), <second> by the low part: tile of
//                          2048 records, one returning LDS atomic per record = its rank in its digit, one global atomic
//                          per (tile, digit), records leave in runs of consecutive 24-byte records
//   group_place_kernel     one workgroup per group: the rows' 64-bit hit masks in LDS (ds_or), their prefix sum, then
//                          every record goes to start(group) + prefix(row) + popcount(mask bits below it) with its
//                          (container, from0InProt) filled in from the row's geometry record; also writes off[row],
//                          from which container_hit_start[] is read
// A chunk of the batch is a contiguous range of rows and its hits a contiguous range of hits[] that starts at *base
// (chained on the device by chunk_base_kernel), as before.
#pragma once

#include "kg_device.hpp"
#include "kg_partition.hpp"

namespace kg {

constexpr uint32_t kHThreads = 512;
constexpr uint32_t kHPer = 4;                       // records per thread and tile
constexpr uint32_t kHTile = kHThreads * kHPer;      // 2048 records = 48 KB of LDS
constexpr uint32_t kHDigits = 128;                  // groups per first-level bucket
constexpr uint32_t kMaxGroups = kHDigits * kHDigits;

// per-row geometry: what turns (row, olane) back into (container, from0InProt)
struct RowGeo { uint32_t container; int32_t pos_first; };      // from0InProt of olane 0

// One record per row of the batch, written once per scan in block-major order (six coalesced streams).
template <bool AA>
__global__ void row_geo_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks, RowGeo *__restrict__ geo)
{
    constexpr uint32_t ROWS = AA ? 1 : 6;
    const uint64_t tl = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tl >= (uint64_t)n_blocks * ROWS) return;
    // thread -> (row kind r, block it): consecutive threads, consecutive blocks of one kind: consecutive rows of a container
    const uint32_t r = (uint32_t)(tl / n_blocks), it = (uint32_t)(tl % n_blocks);
    const BlockDesc bd = blocks[it];
    RowGeo g;
    int32_t pos0;
    row_record_key<AA>(bd, (int)r, 0, &g.container, &pos0);
    g.pos_first = (!AA && r >= 3) ? pos0 - 63 : pos0;            // '-' rows: positions descend with the lane (olane = 63 - lane)
    geo[row_index<AA>(bd, it, (int)r)] = g;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kHThreads) void hit_hist_kernel(const kg_hit *__restrict__ ulist, const uint32_t *__restrict__ chunk_used,
                                                             const unsigned long long *__restrict__ cursor, uint64_t ulist_cap,
                                                             uint32_t g0, uint32_t gs /* 6 + gshift */, uint32_t n_groups,
                                                             uint32_t *ghist /* [n_groups], zeroed */)
{
    extern __shared__ uint32_t h_lds[];
    for (uint32_t g = threadIdx.x; g < n_groups; g += kHThreads) h_lds[g] = 0;
    __syncthreads();
    const unsigned long long cur = *cursor;
    const uint32_t n_chunks = (uint32_t)((cur < ulist_cap ? cur : ulist_cap) / kUChunk);
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t used = chunk_used[c];
        for (uint32_t k = threadIdx.x; k < used; k += kHThreads) {
            const uint32_t g = (ulist[(uint64_t)c * kUChunk + k].container >> gs) - g0;
            if (g < n_groups) atomicAdd(&h_lds[g], 1u);
        }
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < n_groups; g += kHThreads)
        if (h_lds[g]) atomicAdd(&ghist[g], h_lds[g]);
}

// single workgroup: gbase[0 .. n_groups] = exclusive scan of ghist (gbase[n_groups] = the chunk's hits);
// cur1[d] = start of first-level bucket d, cur2[g] = gbase[g]: the cursors the two partition passes draw from;
// tile_start[d] = tiles of kHTile records in the first-level buckets before d (the second pass's work items), d = 0 .. kHDigits.
// Four waves only: a 16-wave workgroup finds no CU with 16 free wave slots while a tag pass (16 per CU) and the ordering
// kernels of another chunk are resident, and waited for the end of the tag pass (2.1 ms, profiles/r03_ordering.md).
constexpr uint32_t kGsThreads = 256, kGsPer = 4;
__global__ __launch_bounds__(kGsThreads) void group_scan_kernel(const uint32_t *__restrict__ ghist, uint32_t n_groups, uint32_t *__restrict__ gbase,
                                                                uint32_t *__restrict__ cur1, uint32_t *__restrict__ cur2, uint64_t *total_out,
                                                                uint32_t *__restrict__ tile_start)
{
    __shared__ uint32_t wsum[kGsThreads / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = 0; b < n_groups; b += kGsThreads * kGsPer) {
        const uint32_t i0 = b + threadIdx.x * kGsPer;              // kGsPer consecutive groups per thread
        uint32_t x[kGsPer], mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < kGsPer; k++) {
            x[k] = i0 + k < n_groups ? ghist[i0 + k] : 0u;
            mine += x[k];
        }
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, all = 0;
        for (int w = 0; w < (int)(kGsThreads / 64); w++) {
            const uint32_t sw = wsum[w];
            if (w < wave) wbase += sw;
            all += sw;
        }
        const uint32_t c = carry;
        uint32_t e = c + wbase + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < kGsPer; k++) {
            const uint32_t i = i0 + k;
            if (i < n_groups) {
                gbase[i] = e;
                cur2[i] = e;
                if ((i & (kHDigits - 1)) == 0) cur1[i / kHDigits] = e;
            }
            e += x[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = c + all;
        __syncthreads();
    }
    const uint32_t n_d1 = (n_groups + kHDigits - 1) / kHDigits;
    if (threadIdx.x == 0) {
        gbase[n_groups] = carry;
        cur1[n_d1] = carry;                                      // end sentinel of the last first-level bucket
        *total_out = carry;
    }
    __syncthreads();                                             // cur1[0 .. n_d1] as written above, by this workgroup
    const uint32_t d = threadIdx.x;                              // (two waves' worth of buckets; the others carry zeros)
    const uint32_t t = d < n_d1 ? (cur1[d + 1] - cur1[d] + kHTile - 1) / kHTile : 0u;
    uint32_t incl = t;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (d == 63) wsum[0] = incl;
    __syncthreads();
    if (d < kHDigits) {
        const uint32_t before = d >= 64 ? wsum[0] : 0u;
        tile_start[d] = before + incl - t;
        if (d == kHDigits - 1) tile_start[kHDigits] = before + incl;
    }
}

// One partition pass.  FIRST: input = the unordered list in its reservation chunks (kUChunk slots, chunk_used filled),
// digit = group / kHDigits, cursors cur1[]; else: input = the first pass's output, bucket d1 = [seg[d1], seg[d1 + 1]) with
// seg[] = the group starts at multiples of kHDigits (gbase), digit = group % kHDigits, cursors cur2[d1 * kHDigits + digit].
template <bool FIRST>
__global__ __launch_bounds__(kHThreads) void hit_partition_kernel(const kg_hit *__restrict__ in, const uint32_t *__restrict__ chunk_used,
                                                                  const unsigned long long *__restrict__ cursor, uint64_t in_cap,
                                                                  const uint32_t *__restrict__ gbase, uint32_t n_groups, uint32_t g0, uint32_t gs,
                                                                  uint32_t *cur, kg_hit *__restrict__ out, uint64_t out_cap,
                                                                  const uint32_t *__restrict__ tile_start)
{
    __shared__ __attribute__((aligned(16))) kg_hit stage[kHTile];
    __shared__ uint32_t hist[kHDigits], cnt[kHDigits], sbase[kHDigits], gb[kHDigits];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < kHDigits) hist[tid] = 0;
    const uint32_t n_d1 = (n_groups + kHDigits - 1) / kHDigits;
    uint32_t n_res = 0;
    if (FIRST) {
        const unsigned long long cc = *cursor;
        n_res = (uint32_t)((cc < in_cap ? cc : in_cap) / kUChunk);
    }
    // work items: FIRST: tiles of kHTile / kUChunk reservation chunks; else the tiles of all first-level buckets, numbered
    // through (tstart[d1] = tiles of the buckets before d1, from group_scan_kernel), so that every workgroup has work whatever
    // the bucket sizes.  (One thread summing the buckets here -- every workgroup reading the same 2 x n_d1 words one after the
    // other -- cost 0.4 ms per launch: 768 requests for one line queue up at its L2 channel.)
    __shared__ uint32_t tstart[kHDigits + 1];
    if (!FIRST) {
        if (tid <= kHDigits) tstart[tid] = tile_start[tid];
        __syncthreads();
    }
    const uint32_t n_items = FIRST ? (n_res + kHTile / kUChunk - 1) / (kHTile / kUChunk) : tstart[kHDigits];
    {
        for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
            uint32_t d1 = 0, tile = item;
            if (!FIRST) {
                uint32_t lo_d = 0, hi_d = n_d1;                      // largest d with tstart[d] <= item (buckets without tiles are skipped)
                while (hi_d - lo_d > 1) {
                    const uint32_t mid = (lo_d + hi_d) / 2;
                    if (tstart[mid] <= item) lo_d = mid; else hi_d = mid;
                }
                d1 = lo_d;
                tile = item - tstart[d1];
            }
            const uint32_t seg_lo = FIRST ? 0u : gbase[d1 * kHDigits];
            const uint32_t seg_hi = FIRST ? 0u : gbase[min((d1 + 1) * kHDigits, n_groups)];
            kg_hit rec[kHPer];
            uint32_t dr[kHPer];
#pragma unroll
            for (uint32_t k = 0; k < kHPer; k++) {
                const uint32_t i = k * kHThreads + tid;            // slot inside the tile
                bool ok;
                uint64_t at;
                if (FIRST) {
                    const uint32_t c = tile * (kHTile / kUChunk) + i / kUChunk;
                    ok = c < n_res && (i % kUChunk) < chunk_used[c];
                    at = (uint64_t)c * kUChunk + (i % kUChunk);
                } else {
                    at = (uint64_t)seg_lo + (uint64_t)tile * kHTile + i;
                    ok = at < seg_hi;
                }
                dr[k] = ~0u;
                if (ok) {
                    rec[k] = stream_load_hit(in + at);
                    const uint32_t g = (rec[k].container >> gs) - g0;
                    if (g < n_groups) {                              // (always; a record outside would be a kernel bug: dropped)
                        const uint32_t d = FIRST ? g / kHDigits : g % kHDigits;
                        dr[k] = (d << 16) | atomicAdd(&hist[d], 1u);
                    }
                }
            }
            __syncthreads();
            if (tid < kHDigits) {                                    // two waves: scan of 128 counts, one global atomic per digit
                const uint32_t c = hist[tid];
                uint32_t incl = c;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t y = __shfl_up(incl, off);
                    if ((int)lane >= off) incl += y;
                }
                cnt[tid] = c;
                sbase[tid] = incl - c;                               // (second wave: + the first wave's total, below)
                gb[tid] = c ? atomicAdd(&cur[FIRST ? tid : d1 * kHDigits + tid], c) : 0u;
                hist[tid] = 0;
            }
            __syncthreads();
            const uint32_t first_half = sbase[63] + cnt[63];
#pragma unroll
            for (uint32_t k = 0; k < kHPer; k++)
                if (dr[k] != ~0u) {
                    const uint32_t d = dr[k] >> 16;
                    stage[sbase[d] + (d >= 64 ? first_half : 0u) + (dr[k] & 0xFFFFu)] = rec[k];
                }
            __syncthreads();
            // a digit's run leaves as a flat stream of 8-byte words (three per record): consecutive lanes, consecutive words, so
            // every store instruction covers whole lines (record-wise 24-byte stores wrote every line of a run three times in
            // thirds and the partition passes' WRITE_SIZE was twice their data)
            const uint64_t *stage64 = reinterpret_cast<const uint64_t *>(stage);
            uint64_t *out64 = reinterpret_cast<uint64_t *>(out);
            for (uint32_t d = wave; d < kHDigits; d += kHThreads / 64u) {
                const uint32_t n = cnt[d];
                if (n == 0) continue;
                const uint32_t s0 = sbase[d] + (d >= 64 ? first_half : 0u);
                const uint64_t o0 = gb[d];
                const uint32_t n_ok = o0 >= out_cap ? 0u : (uint32_t)min((uint64_t)n, out_cap - o0);
                for (uint32_t i = lane; i < 3u * n_ok; i += 64u) stream_store8(out64 + o0 * 3u + i, stage64[(uint64_t)s0 * 3u + i]);
            }
            __syncthreads();                                         // stage / cnt / sbase / gb are rewritten by the next tile
        }
    }
}

// (One pass instead -- every record straight to its group's range through a cursor per group, the write-back L2 combining the
//  24-byte stores -- was built and measured in round 4: byte-identical, stage 15.5 -> 15.95 ms, 125 Mbp shard 2.99 -> 3.19:
//  profiles/r04_experiments.md.)
// One workgroup per group of 2^gshift rows.  Dynamic LDS: (8 + 4) << gshift bytes, and with `staged` (8 + 4 + 8) << gshift
// + kPlaceOut * 24: the rows' geometry records are brought in with one coalesced read and the group's records are put in
// order in LDS and leave as a flat stream of 8-byte words.  (Placed straight from the registers every record cost a gather
// lane for its row's geometry and two store lanes into the group's output range, 64 different lines per wave
// instruction: 110 M partially used lines per Gbp in the CUs' vector memory path, which is what the stage's passes compete
// for.)  A group with more than kPlaceOut records (dense inputs) is placed directly.
constexpr uint32_t kPlaceOut = 1536;       // a group of 1024 rows holds ~1150 hits of the 1 Gbp contig mix
__host__ __device__ inline size_t group_place_lds(uint32_t gshift, bool staged)
{
    return staged ? ((size_t)20 << gshift) + (size_t)kPlaceOut * sizeof(kg_hit) : ((size_t)12 << gshift);
}
template <bool AA>
__global__ __launch_bounds__(kHThreads) void group_place_kernel(const kg_hit *__restrict__ in, const uint32_t *__restrict__ gbase, uint32_t n_groups,
                                                                uint32_t g0, uint32_t gshift, uint32_t row_lo, uint32_t row_hi /* the chunk's rows */,
                                                                const RowGeo *__restrict__ geo, uint64_t n_rows_all, uint32_t staged,
                                                                const uint64_t *__restrict__ base,
                                                                kg_hit *__restrict__ hits, uint64_t hits_cap, uint32_t *__restrict__ offs,
                                                                uint32_t *__restrict__ hit_slots /* KG_F_PROGRESS: parallel to hits[], else null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char p_lds[];
    const uint32_t R = 1u << gshift;
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(p_lds);
    uint32_t *pre = reinterpret_cast<uint32_t *>(masks + R);
    RowGeo *lgeo = reinterpret_cast<RowGeo *>(pre + R);                   // (staged only)
    uint64_t *lout = reinterpret_cast<uint64_t *>(lgeo + R);              // (staged only) kPlaceOut records of three words
    __shared__ uint32_t wsum[kHThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t chunk_base = *base;
    const uint32_t per = R / kHThreads > 0 ? R / kHThreads : 1u;        // rows per thread in the scan (R >= kHThreads, or some threads idle)
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t lo = gbase[g], n = gbase[g + 1] - lo;
        const uint64_t row0 = (uint64_t)(g0 + g) << gshift;
        const bool to_lds = staged && n <= kPlaceOut;
        for (uint32_t i = tid; i < R; i += kHThreads) {
            masks[i] = 0ull;
            if (staged && n && row0 + i < n_rows_all) lgeo[i] = geo[row0 + i];
        }
        __syncthreads();
        // the group's records are read once and wait in registers for their rank when there are at most kKeep per thread
        // (a group of 1024 rows holds ~1150 hits of the 1 Gbp contig mix); larger groups are read a second time
        constexpr uint32_t kKeep = 4;
        const bool keep = n <= kHThreads * kKeep;
        kg_hit rk[kKeep];
        if (keep) {
#pragma unroll
            for (uint32_t j = 0; j < kKeep; j++) {
                const uint32_t k = j * kHThreads + tid;
                if (k < n) {
                    rk[j] = stream_load_hit(in + lo + k);
                    atomicOr(&masks[(rk[j].container >> 6) & (R - 1u)], 1ull << (rk[j].container & 63u));
                }
            }
        } else {
            for (uint32_t k = tid; k < n; k += kHThreads) {
                const uint32_t key = in[lo + k].container;
                atomicOr(&masks[(key >> 6) & (R - 1u)], 1ull << (key & 63u));
            }
        }
        __syncthreads();
        // exclusive prefix of the rows' hit counts: `per` consecutive rows per thread, then a workgroup scan
        uint32_t mine = 0;
        const uint32_t r_lo = tid * per;
        if (r_lo < R)
            for (uint32_t i = 0; i < per; i++) mine += (uint32_t)__popcll(masks[r_lo + i]);
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off);
            if ((int)lane >= off) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; w++) run += wsum[w];
        if (r_lo < R)
            for (uint32_t i = 0; i < per; i++) {
                pre[r_lo + i] = run;
                const uint64_t row = row0 + r_lo + i;
                if (row >= row_lo && row < row_hi) offs[row] = (uint32_t)(chunk_base + lo + run);
                run += (uint32_t)__popcll(masks[r_lo + i]);
            }
        __syncthreads();
        auto place = [&](kg_hit h) {
            const uint32_t key = h.container, rl = (key >> 6) & (R - 1u), ol = key & 63u;
            const RowGeo gr = staged ? lgeo[rl] : geo[key >> 6];
            const uint32_t rank = pre[rl] + (uint32_t)__popcll(masks[rl] & ((1ull << ol) - 1ull));      // inside the group
            if (hit_slots && chunk_base + lo + rank < hits_cap)
                hit_slots[chunk_base + lo + rank] = (uint32_t)h.from0InProt;      // the verify pass left the found slot there
            h.container = gr.container;
            h.from0InProt = gr.pos_first + (int32_t)ol;
            if (to_lds) {
                uint64_t w3[3];
                __builtin_memcpy(w3, &h, 24);
                lout[3u * rank] = w3[0]; lout[3u * rank + 1] = w3[1]; lout[3u * rank + 2] = w3[2];
            } else {
                const uint64_t dst = chunk_base + lo + rank;
                if (dst < hits_cap) hits[dst] = h;          // only out of range when a list overflowed: that scan is re-run
            }
        };
        if (keep) {
#pragma unroll
            for (uint32_t j = 0; j < kKeep; j++)
                if (j * kHThreads + tid < n) place(rk[j]);
        } else {
            for (uint32_t k = tid; k < n; k += kHThreads) place(in[lo + k]);
        }
        __syncthreads();                                    // masks / pre / lgeo are rewritten by the next group; lout is complete
        if (to_lds) {
            const uint64_t dst0 = chunk_base + lo;
            const uint32_t n_ok = dst0 >= hits_cap ? 0u : (uint32_t)min((uint64_t)n, hits_cap - dst0);
            uint64_t *out64 = reinterpret_cast<uint64_t *>(hits) + dst0 * 3u;
            for (uint32_t i = tid; i < 3u * n_ok; i += kHThreads) stream_store8(out64 + i, lout[i]);
            __syncthreads();                                // lout is rewritten by the next group
        }
    }
}

// ---------------------------------------------------------------------------------------
// Multi-GPU exchange (kmergutsjava_amd/distributed.py): the hit records a rank sends are ordered by ITS containers; on the
// gathering rank the records of local sequence k (one contiguous piece: a sequence's containers are adjacent) move to
// dst_first[k] .. with container += container_shift[k].  A segmented copy: one binary search per record over the local
// sequence starts (L2-resident), 24 bytes in, 24 bytes out, both coalesced.
__global__ __launch_bounds__(256) void restore_hits_kernel(const kg_hit *__restrict__ src, uint64_t n_hits, const int64_t *__restrict__ seq_first,
                                                           uint64_t n_seqs, const int64_t *__restrict__ dst_first,
                                                           const int32_t *__restrict__ container_shift, kg_hit *__restrict__ dst)
{
    // a workgroup takes 1024 consecutive records at a time; the sequence of the first one is found by ONE binary search per
    // tile, every thread then steps forward from there (sequences are thousands of records long; short reads: a few steps)
    __shared__ uint64_t k_first;
    constexpr uint64_t kTile = 1024;
    const uint64_t n_tiles = (n_hits + kTile - 1) / kTile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t i0 = tile * kTile;
            uint64_t lo = 0, hi = n_seqs;                   // largest k with seq_first[k] <= i0  (seq_first[n_seqs] == n_hits > i0)
            while (hi - lo > 1) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if ((uint64_t)seq_first[mid] <= i0) lo = mid; else hi = mid;
            }
            k_first = lo;
        }
        __syncthreads();
        uint64_t k = k_first;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t i = tile * kTile + (uint64_t)j * 256 + threadIdx.x;
            if (i < n_hits) {
                while (k + 1 < n_seqs && (uint64_t)seq_first[k + 1] <= i) k++;
                kg_hit h = src[i];
                h.container += (uint32_t)container_shift[k];
                dst[(uint64_t)dst_first[k] + (i - (uint64_t)seq_first[k])] = h;
            }
        }
    }
}

}  // namespace kg
