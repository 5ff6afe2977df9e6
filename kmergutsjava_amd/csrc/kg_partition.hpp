// kg_partition.hpp -- the partitioned scan: same results as scan_kernel, an order of magnitude less DRAM traffic.
//
// Why: a probe is a random 16-byte read; from a table far larger than the 4 MiB L2 of an XCD every probe
// costs a whole 128-byte line from the memory side and the chip tops out at ~50 G such reads per second
// (profiles/r01_gather_ceiling*.jsonl), while L2-resident random reads run at 220-270 G/s.  So the query
// k-mers are first bucketed by slot range (a bucket's tag range <= 2 MiB), then probed bucket by bucket with
// all the workgroups of one XCD working on the same bucket.  The reference does the same thing for the same
// reason with a sort and a sequential merge (KGJ:1076-1095, 944-1034); here it is one counting pass, one
// scatter pass (8 bytes per query, written and read once, sequentially) and the probe pass.
//
//   part_kernel<COUNT>   : encode every window; per-wave histogram over buckets                (no data moved)
//   part_offsets_kernel  : exclusive scan per bucket over the waves + bucket starts
//   part_kernel<SCATTER> : encode again; entry -> its wave's private region of its bucket (no global atomics,
//                          deterministic layout, exact sizes)
//   bucket_probe_kernel  : persistent workgroups; group x = blockIdx % 8 (XCD under round-robin placement,
//                          speed only) walks the buckets b % 8 == x; 16-tag probe as in probe_n; a hit sets bit
//                          `lane` in the 64-bit mask of its (block,row) and appends {id, payload} to an
//                          unordered list
//   rows_from_masks      : popcount of the masks -> counts[] in container-major row order (then the usual
//                          prefix sum)
//   place_unordered      : unordered list -> hits[] at offs[row] + popcount(mask bits before the lane)
//
// Entry (64 bit): low word = quotient << shift | slot_low, high word = id = block << 9 | row << 6 | lane;
// value = quotient * numSigs + (bucket << shift | slot_low) exactly.
#pragma once

#include "kg_device.hpp"

namespace kg {

constexpr int kMaxBuckets = 1024;
constexpr int kProbeN = 4;                  // queries per lane per iteration of the bucket probe
constexpr uint32_t kUChunk = 512;           // records per reservation of the unordered hit list

// exact quotient and remainder of v by num_sigs (see home_slot)
__device__ __forceinline__ uint64_t home_slot_q(uint64_t v, uint64_t num_sigs, uint64_t magic, uint32_t *q_out)
{
    uint64_t q = __umul64hi(v, magic);
    uint64_t r = v - q * num_sigs;
    if (r >= num_sigs) { r -= num_sigs; q += 1; }
    *q_out = (uint32_t)q;
    return r;
}

constexpr uint64_t kEntInvalid = ~0ull;     // filler of the padded tail of a (wave, bucket) region

template <bool AA>
inline size_t part_lds_bytes(bool scatter, uint32_t n_buckets)
{
    size_t enc = (sizeof(typename WaveLds<AA>::type) + 15) & ~(size_t)15;
    return enc + (scatter ? (size_t)n_buckets * 32 + (size_t)n_buckets * 8 : (size_t)n_buckets * 4);
}

// One wave per workgroup (the wave's LDS: encode scratch, per-bucket cursor, and in the scatter pass a
// 4-entry = 32-byte write-combining buffer per bucket).  Regions of one (wave, bucket) pair are padded to a
// multiple of 4 entries so that every flush is one aligned 32-byte sector; the padding carries kEntInvalid.
// Two lanes of one step that target the same bucket are serialised by a claim word (last writer wins, the
// others retry), so the buffers need no atomics.
template <bool AA, bool SCATTER>
__global__ __launch_bounds__(kWave) void part_kernel(
    const uint8_t *__restrict__ seq, const BlockDesc *__restrict__ blocks, uint32_t n_blocks, uint64_t limit,
    uint64_t num_sigs, uint64_t magic, uint32_t shift, uint32_t n_buckets, uint32_t *__restrict__ M /* [bucket][wave] */,
    const uint32_t *__restrict__ bstart, uint64_t *__restrict__ ent, unsigned long long *ctr)
{
    constexpr int ROWS = AA ? 1 : 6;
    // dynamic LDS (part_lds_bytes): encode scratch | [SCATTER: 4-entry buffers] | cursors | [SCATTER: claims]
    extern __shared__ __attribute__((aligned(16))) unsigned char part_lds[];
    typedef typename WaveLds<AA>::type Enc;
    Enc &l = *reinterpret_cast<Enc *>(part_lds);
    constexpr size_t enc_bytes = (sizeof(Enc) + 15) & ~(size_t)15;
    uint64_t *buf = reinterpret_cast<uint64_t *>(part_lds + enc_bytes);
    uint32_t *wpos = reinterpret_cast<uint32_t *>(part_lds + enc_bytes + (SCATTER ? (size_t)n_buckets * 32 : 0));
    uint32_t *claim = wpos + n_buckets;                                     // COUNT: unused
    const int lane = threadIdx.x;
    const uint32_t wave_global = blockIdx.x;
    const uint32_t n_waves = gridDim.x;
    for (uint32_t b = lane; b < n_buckets; b += 64) {
        wpos[b] = SCATTER ? M[(uint64_t)b * n_waves + wave_global] + bstart[b] : 0u;
        if (SCATTER) { buf[4 * b] = kEntInvalid; buf[4 * b + 1] = kEntInvalid; buf[4 * b + 2] = kEntInvalid; buf[4 * b + 3] = kEntInvalid; }
    }
    encode_init<AA>(l, lane);          // ends with a wave_sync

    unsigned long long n_valid = 0;
    for (uint32_t it = wave_global; it < n_blocks; it += n_waves) {
        const BlockDesc bd = blocks[it];
        encode_block<AA>(l, seq, bd, lane);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            uint64_t v;
            bool valid = row_value<AA>(l, r, lane, bd, &v);
            uint32_t q;
            const uint64_t slot = home_slot_q(v, num_sigs, magic, &q);
            if (!SCATTER && valid) n_valid++;                       // query k-mers (KGJ:913-920)
            valid = valid && slot < limit;                          // beyond the stream: never probed
            const uint32_t b = (uint32_t)(slot >> shift);
            if (!SCATTER) {
                if (valid) atomicAdd(&wpos[b], 1u);
            } else {
                const uint32_t low = (q << shift) | ((uint32_t)slot & ((1u << shift) - 1u));
                const uint32_t id = (it << 9) | ((uint32_t)r << 6) | (uint32_t)lane;
                const uint64_t e = ((uint64_t)id << 32) | low;
                bool pending = valid;
                while (__ballot(pending)) {
                    if (pending) claim[b] = (uint32_t)lane;
                    wave_sync();
                    if (pending && claim[b] == (uint32_t)lane) {     // sole owner of bucket b in this round
                        const uint32_t at = wpos[b];
                        buf[4 * b + (at & 3u)] = e;
                        wpos[b] = at + 1;
                        if ((at & 3u) == 3u) {                       // sector complete: one aligned 32-byte write
                            const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(&buf[4 * b]);
                            const ulonglong2 hi = *reinterpret_cast<const ulonglong2 *>(&buf[4 * b + 2]);
                            ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(ent + (at - 3u));
                            dst[0] = lo; dst[1] = hi;
                            const ulonglong2 inv = make_ulonglong2(kEntInvalid, kEntInvalid);
                            *reinterpret_cast<ulonglong2 *>(&buf[4 * b]) = inv;
                            *reinterpret_cast<ulonglong2 *>(&buf[4 * b + 2]) = inv;
                        }
                        pending = false;
                    }
                    wave_sync();
                }
            }
        }
        wave_sync();   // LDS is reused by the next block
    }
    if (!SCATTER) {
        // padded to a multiple of 4 entries (see above)
        for (uint32_t b = lane; b < n_buckets; b += 64) M[(uint64_t)b * n_waves + wave_global] = (wpos[b] + 3u) & ~3u;
        for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off);
        if (lane == 0) atomicAdd(&ctr[0], n_valid);
    } else {
        // partial last sectors, with their fillers
        for (uint32_t b = lane; b < n_buckets; b += 64) {
            const uint32_t at = wpos[b];
            if (at & 3u) {
                ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(ent + (at & ~3u));
                dst[0] = *reinterpret_cast<const ulonglong2 *>(&buf[4 * b]);
                dst[1] = *reinterpret_cast<const ulonglong2 *>(&buf[4 * b + 2]);
            }
        }
    }
}

// one workgroup per bucket: exclusive scan of the bucket's per-wave counts in place, total to tot[b]
__global__ __launch_bounds__(256) void part_offsets_kernel(uint32_t *__restrict__ M, uint32_t n_waves, uint32_t *__restrict__ tot)
{
    __shared__ uint32_t lds[8];
    __shared__ uint32_t carry;
    uint32_t *row = M + (uint64_t)blockIdx.x * n_waves;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_waves; base += 256 * 8) {
        uint32_t v[8], s = 0;
        const uint32_t i0 = base + threadIdx.x * 8;
        for (int k = 0; k < 8; k++) { v[k] = i0 + k < n_waves ? row[i0 + k] : 0; s += v[k]; }
        uint32_t total;
        uint32_t run = carry + wg_exclusive_scan(s, &total, lds);
        for (int k = 0; k < 8; k++) { if (i0 + k < n_waves) row[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// single workgroup: bstart[b] = sum of tot[0..b), bstart[n] = total
__global__ __launch_bounds__(256) void part_bstart_kernel(const uint32_t *__restrict__ tot, uint32_t n_buckets,
                                                          uint32_t *__restrict__ bstart, uint64_t *total_out)
{
    __shared__ uint32_t lds[8];
    uint32_t v[4], s = 0;
    const uint32_t i0 = threadIdx.x * 4;
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < n_buckets ? tot[i0 + k] : 0; s += v[k]; }
    uint32_t total;
    uint32_t run = wg_exclusive_scan(s, &total, lds);
    for (int k = 0; k < 4; k++) { if (i0 + k <= n_buckets) bstart[i0 + k] = run; run += v[k]; }
    if (threadIdx.x == 0) *total_out = total;
}

// Work distribution: the workgroups with blockIdx % 8 == x (one XCD under round-robin placement -- measured,
// tools/xcd_affinity.hip; speed only, never correctness) walk the buckets b % 8 == x in order and take chunks
// of the current bucket from a per-bucket counter, so an XCD's L2 holds one bucket's tags (two at a hand-over).
template <bool AA, bool COUNTERS>
__global__ __launch_bounds__(256) void bucket_probe_kernel(
    const uint8_t *__restrict__ entries, const uint8_t *__restrict__ tags, uint64_t limit, uint64_t num_sigs, uint64_t magic,
    const uint64_t *__restrict__ ent, const uint32_t *__restrict__ bstart, uint32_t n_buckets, uint32_t shift,
    uint32_t *next_chunk /* [n_buckets], zeroed */, kg_hit *__restrict__ ulist, uint32_t *__restrict__ chunk_used,
    unsigned long long *cursor, uint64_t ulist_cap, unsigned long long *__restrict__ masks, unsigned long long *ctr)
{
    constexpr int N = kProbeN;
    constexpr uint32_t ROWS = AA ? 1 : 6;
    constexpr uint32_t kChunk = 256u * N * 2u;         // entries per grab
    __shared__ uint32_t s_chunk;
    const int lane = threadIdx.x & 63;
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic;
    unsigned long long ctr_dummy = 0, ctr_slots = 0;
    unsigned long long cur_base = 0;                   // this wave's reservation in the unordered list (uniform)
    uint32_t cur_used = kUChunk;                       // "full": the first append takes a chunk
    bool have_chunk = false;

    for (uint32_t b = blockIdx.x & 7u; b < n_buckets; b += 8) {
        const uint32_t lo = bstart[b], hi = bstart[b + 1];
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_chunk = atomicAdd(&next_chunk[b], 1u);
            __syncthreads();
            const uint64_t c_lo = (uint64_t)lo + (uint64_t)s_chunk * kChunk;
            if (c_lo >= hi) break;                      // bucket exhausted (uniform): next bucket of this group
            for (uint32_t c0 = (uint32_t)c_lo; c0 < hi && c0 < c_lo + kChunk; c0 += 256u * N) {
                uint64_t val[N];
                bool valid[N];
                uint32_t id[N];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const uint32_t i = c0 + (uint32_t)k * 256u + threadIdx.x;
                    const uint64_t e = i < hi ? ent[i] : kEntInvalid;
                    valid[k] = e != kEntInvalid;
                    const uint32_t low = (uint32_t)e;
                    id[k] = (uint32_t)(e >> 32);
                    const uint64_t slot = ((uint64_t)b << shift) | (low & ((1u << shift) - 1u));
                    val[k] = (uint64_t)(low >> shift) * num_sigs + slot;
                }
                Payload pay[N];
                const uint32_t foundm = probe_n<N, COUNTERS>(tab, val, valid, pay, ctr_dummy, ctr_slots);

                // hits: set the lane's bit in the (block,row) mask; append {id, payload} to the unordered list
                uint32_t cnt[N], rank[N], total = 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const bool f = (foundm >> k) & 1u;
                    const unsigned long long m = __ballot(f);
                    cnt[k] = (uint32_t)__popcll(m);
                    rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    total += cnt[k];
                    if (f) atomicOr(&masks[(uint64_t)(id[k] >> 9) * ROWS + ((id[k] >> 6) & 7u)], 1ull << (id[k] & 63u));
                }
                if (total) {
                    if (cur_used + total > kUChunk) {               // uniform: retire the chunk, take a new one
                        if (lane == 0 && have_chunk && cur_base + kUChunk <= ulist_cap) chunk_used[cur_base / kUChunk] = cur_used;
                        unsigned long long nb = 0;
                        if (lane == 0) nb = atomicAdd(cursor, (unsigned long long)kUChunk);
                        nb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) |
                             (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)nb);
                        cur_base = nb; cur_used = 0; have_chunk = true;
                    }
                    if (cur_base + kUChunk <= ulist_cap) {
                        uint32_t at = cur_used;
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            if ((foundm >> k) & 1u) {
                                kg_hit h;
                                h.container = id[k];
                                h.from0InProt = 0;
                                h.oI = pay[k].oI; h.avgOffFromEnd = pay[k].avg; h.fI = pay[k].fI; h.functionWt = pay[k].wt;
                                ulist[cur_base + at + rank[k]] = h;
                            }
                            at += cnt[k];
                        }
                    }
                    cur_used += total;
                }
            }
        }
    }
    if (lane == 0 && have_chunk && cur_base + kUChunk <= ulist_cap) chunk_used[cur_base / kUChunk] = cur_used;
    if (COUNTERS) {
        for (int off = 32; off > 0; off >>= 1) ctr_slots += __shfl_down(ctr_slots, off);
        if (lane == 0) atomicAdd(&ctr[1], ctr_slots);
    }
}

// counts[row] = number of hits of (block,row): one thread per (block,row)
template <bool AA>
__global__ void rows_from_masks_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks,
                                       const unsigned long long *__restrict__ masks, uint32_t *__restrict__ counts)
{
    constexpr uint32_t ROWS = AA ? 1 : 6;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_blocks * ROWS) return;
    const uint32_t it = (uint32_t)(t / ROWS), r = (uint32_t)(t % ROWS);
    const BlockDesc bd = blocks[it];
    counts[row_index<AA>(bd, it, (int)r)] = (uint32_t)__popcll(masks[t]);
}

// unordered list -> hits[] ordered by (container, from0InProt); one workgroup per reservation chunk
template <bool AA>
__global__ __launch_bounds__(256) void place_unordered_kernel(const BlockDesc *__restrict__ blocks,
                                                              const unsigned long long *__restrict__ masks,
                                                              const uint32_t *__restrict__ offs,
                                                              const kg_hit *__restrict__ ulist,
                                                              const uint32_t *__restrict__ chunk_used, uint32_t n_chunks,
                                                              kg_hit *__restrict__ hits)
{
    constexpr uint32_t ROWS = AA ? 1 : 6;
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const uint32_t used = chunk_used[c];
    for (uint32_t k = threadIdx.x; k < used; k += blockDim.x) {
        kg_hit h = ulist[(uint64_t)c * kUChunk + k];
        const uint32_t id = h.container;
        const uint32_t it = id >> 9, r = (id >> 6) & 7u, ln = id & 63u;
        const BlockDesc bd = blocks[it];
        const unsigned long long m = masks[(uint64_t)it * ROWS + r];
        // '+' rows ascend with the lane, '-' rows descend (see scan_kernel)
        const unsigned long long before = (!AA && r >= 3) ? (ln == 63 ? 0ull : (m >> (ln + 1))) : (m & ((1ull << ln) - 1ull));
        row_record_key<AA>(bd, (int)r, (int)ln, &h.container, &h.from0InProt);
        hits[(uint64_t)offs[row_index<AA>(bd, it, (int)r)] + (uint32_t)__popcll(before)] = h;
    }
}

}  // namespace kg
