// kg_partition.hpp -- the partitioned scan: same results as scan_kernel, an order of magnitude less DRAM traffic.
//
// Why: a probe is a random 16-byte read; from a table far larger than the 4 MiB L2 of an XCD every probe
// costs a whole 128-byte line from the memory side and the chip tops out at ~50 G such reads per second
// (profiles/r01_gather_ceiling*.jsonl), while L2-resident random reads run at 220-270 G/s (one 64-byte sector
// fill each: the XCD-resident gather bandwidth of MI355X_MICROARCH.md).  So the query
// k-mers are first bucketed by slot range (a bucket's tag range <= 2 MiB), then probed bucket by bucket with
// all the workgroups of one XCD working on the same bucket.  The reference does the same thing for the same
// reason with a sort and a sequential merge (KGJ:1076-1095, 944-1034); here it is one scatter pass (8 bytes
// per query, written and read once in 128-byte lines) and the probe pass.
//
//   part_scatter_kernel  : encode every window once; entry -> 16-entry (128-byte) write-combining buffer of its
//                          bucket in the workgroup's LDS -> the workgroup's over-allocated region of the bucket
//                          (no counting pass, no per-entry global atomics, no workgroup barrier in the main loop;
//                          overflow list for skewed inputs)
//   lowc_blocks_kernel   : the blocks the scatter pass set aside (low-complexity sequence: most windows in one or two
//                          buckets): their entries are appended to the regions in whole same-bucket sets
//   bucket_tag_kernel    : persistent workgroups; group x = blockIdx % 8 (XCD under round-robin placement,
//                          speed only) walks the buckets b % 8 == x; per entry one 16-tag window out of L2: an
//                          empty slot ends it, a fingerprint match or an undecided window becomes a 16-byte
//                          candidate record
//   verify_kernel        : one lane per candidate: (continue the tag walk,) fetch the 24-byte record, compare the
//                          key; a hit appends {key, payload} to an unordered list
//   overflow_probe_kernel: the groups the scatter pass could not fit into their regions
//   kg_order.hpp         : unordered list -> hits[] in (container, from0InProt) order: two partition passes by key range
//                          and an in-LDS ranking per group of rows, all streaming
// The batch is processed in chunks of whole sequences; scatter, tag pass and verification + placement of successive
// chunks run on three streams (kmerguts_hip.hip, scan_impl).
//
// Entry (64 bit): low word = quotient << shift | slot_low, high word = id = the window's key (window_key: its rank in the
// final order of the hit records);
// value = quotient * numSigs + (bucket << shift | slot_low) exactly.
#pragma once

#include "kg_device.hpp"

namespace kg {

constexpr int kMaxBuckets = 1024;
#ifndef KG_PROBE_N
#define KG_PROBE_N 4
#endif
// Register budgets.  The scatter pass of chunk c + 1 and the tag pass of chunk c share the CUs, and what decides how many tag
// waves run beside a scatter workgroup (16 waves, 4 per SIMD) is the SIMD's 512 VGPRs: 4 x 104 (99 rounded to the
// allocation granule of 8) left room for ONE tag wave of 72 (66) per SIMD -- the tag pass ran beside a scatter pass with a
// quarter of its waves.  amdgpu_waves_per_eu(5) holds the scatter kernel to 96 VGPRs (no spill) and the tag kernel's slot
// arithmetic in 32 bits brings it to 61: 4 x 96 + 2 x 64 = 512, two tag waves per SIMD.
#ifndef KG_SCATTER_WPE
#define KG_SCATTER_WPE 5
#endif
#ifndef KG_TAG_WPE
#define KG_TAG_WPE 8
#endif
#if KG_SCATTER_WPE
#define KG_SCATTER_REGS __attribute__((amdgpu_waves_per_eu(KG_SCATTER_WPE)))
#else
#define KG_SCATTER_REGS
#endif
#if KG_TAG_WPE
#define KG_TAG_REGS __attribute__((amdgpu_waves_per_eu(KG_TAG_WPE)))
#else
#define KG_TAG_REGS
#endif
#ifndef KG_STAGE_FLUSH
#define KG_STAGE_FLUSH 64
#endif
// Rows of a DNA block (6: three phases x two strands) that the scatter pass keeps in registers and inserts together.
// 6: all at once (twelve entry registers, six LDS atomics in flight per lane); 3: one strand at a time; 2: a third.
#ifndef KG_SCATTER_RG
#define KG_SCATTER_RG 6
#endif

// 1: the verify pass fetches the record at the home slot whole with the first round of keys (kScanOn candidates): stage
// 16.37 -> 16.20 ms (r04 c04)
#ifndef KG_SCAN_FULL
#define KG_SCAN_FULL 1
#endif

constexpr uint32_t kStageFlush = KG_STAGE_FLUSH;     // candidate records per flush of a tag wave's staging buffer (<= 64)
constexpr int kProbeN = KG_PROBE_N;                  // queries per lane per iteration of the bucket probe
#ifndef KG_INDEX_N
#define KG_INDEX_N KG_PROBE_N
#endif
constexpr int kIndexN = KG_INDEX_N;                  // the same for the byte-index pass (31 VGPRs at 4: room for more in flight)
constexpr uint32_t kUChunk = 512;           // records per reservation of the unordered hit list

constexpr uint64_t kEntInvalid = ~0ull;     // filler entry (padding of a 16-entry group)

// The id an entry carries = the window's KEY = its rank in the final (container, from0InProt) order (kg_order.hpp):
// row_index * 64 + olane; r and bd are wave-uniform.
template <bool AA>
__device__ __forceinline__ uint32_t window_key(const BlockDesc &bd, uint32_t it, int r, int lane)
{
    const uint32_t ol = (!AA && r >= 3) ? 63u - (uint32_t)lane : (uint32_t)lane;
    return (row_index<AA>(bd, it, r) << 6) | ol;
}
#ifndef KG_SCATTER_WAVES
#define KG_SCATTER_WAVES 16
#endif
constexpr int kScatterWaves = KG_SCATTER_WAVES;           // waves per scatter workgroup (one workgroup per CU: its LDS holds the buffers)
constexpr uint32_t kGroup = 16;             // entries per write-combining buffer = one 128-byte line

template <bool AA>
inline size_t scatter_lds_bytes(uint32_t n_buckets)
{
    size_t enc = (sizeof(typename WaveLds<AA>::type) + 15) & ~(size_t)15;
    size_t tab = (sizeof(typename WaveLds<AA>::tables) + 15) & ~(size_t)15;
    return enc * kScatterWaves + tab + (size_t)n_buckets * (kGroup * 8 + 12);
}

// ---------------------------------------------------------------------------------------
// (Storing every entry straight at its place in the region instead -- counters only in LDS, the XCD's write-back L2 as
//  the write-combining buffer: half the VALU, a fifth of the LDS -- was built and measured in round 3: 17.3 ms per Gbp
//  against 7.8 for this kernel; 1.36e9 scattered 8-byte stores cost more than everything they save:
//  profiles/r03_experiments.md.)
// Scatter pass: every window is encoded once; its entry goes into the workgroup's 16-entry buffer of its
// bucket (LDS); a full buffer is written as one aligned 128-byte line into the workgroup's private region of
// that bucket.  Regions are over-allocated (cap entries each, no counting pass); a group that does not fit its
// region goes to the overflow list (skewed inputs), which is probed separately.  Layout of the entry array:
// region (bucket b, workgroup w) = [ (b * n_wg + w) * cap , + fill[b * n_wg + w] ).
// 16 bytes of a flushed group.  KG_SCATTER_NT: with the streaming hint (the entries are read once, a pass later, from HBM)
#ifndef KG_SCATTER_NT
#define KG_SCATTER_NT 1
#endif
__device__ __forceinline__ void store_entry_pair(uint64_t *dst, const ulonglong2 &v)
{
#if KG_SCATTER_NT
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    u64x2 x; x.x = v.x; x.y = v.y;
    __builtin_nontemporal_store(x, reinterpret_cast<u64x2 *>(dst));
#else
    *reinterpret_cast<ulonglong2 *>(dst) = v;
#endif
}

template <bool AA>
__global__ __launch_bounds__(kWave *kScatterWaves) KG_SCATTER_REGS void part_scatter_kernel(
    const uint8_t *__restrict__ seq, const BlockDesc *__restrict__ blocks, uint32_t block_lo, uint32_t n_blocks /* of this launch */,
    uint64_t limit, uint32_t num_sigs /* 64 <= num_sigs < 2^31 */, uint32_t m35, uint32_t shift, uint32_t n_buckets, uint32_t cap,
    uint64_t *__restrict__ ent, uint32_t *__restrict__ fill, uint32_t *ovf_cursor, uint32_t ovf_cap, uint32_t *__restrict__ ovf_bucket,
    uint64_t *__restrict__ ovf_ent, uint32_t *lowc_cursor /* [0] count */, uint32_t *__restrict__ lowc_blocks, unsigned long long *ctr,
    Progress *prog /* KG_F_PROGRESS, else null */, uint32_t insert_prio /* wave priority of the insert phase (0..3) */)
{
    constexpr int ROWS = AA ? 1 : 6;
    typedef typename WaveLds<AA>::type Enc;
    constexpr size_t enc_bytes = (sizeof(Enc) + 15) & ~(size_t)15;
    extern __shared__ __attribute__((aligned(16))) unsigned char part_lds[];
    // (the wave number through readfirstlane: the compiler then knows that the block number, the block descriptor and
    //  everything computed from them -- row indices, the '-' strand's % 3, window keys' row part -- are wave-uniform and
    //  keeps them in scalar registers: 7 quarter-rate multiplies and ~50 VALU per block were per-lane copies of them)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Enc &l = *reinterpret_cast<Enc *>(part_lds + enc_bytes * wave);
    typedef typename WaveLds<AA>::tables Tables;
    constexpr size_t tab_bytes = (sizeof(Tables) + 15) & ~(size_t)15;
    Tables &enc_tables = *reinterpret_cast<Tables *>(part_lds + enc_bytes * kScatterWaves);
    uint64_t *buf = reinterpret_cast<uint64_t *>(part_lds + enc_bytes * kScatterWaves + tab_bytes);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(buf + (size_t)n_buckets * kGroup);
    uint32_t *wrel = cnt + n_buckets;                 // entries already written to this workgroup's region of bucket b
    uint32_t *written = wrel + n_buckets;             // entries of the current group whose LDS store has been issued
    const uint32_t w = blockIdx.x, n_wg = gridDim.x;
    const uint32_t limit32 = limit < 0xFFFFFFFFull ? (uint32_t)limit : 0xFFFFFFFFu;      // slots are < num_sigs < 2^31
    for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x) { cnt[b] = 0; wrel[b] = 0; written[b] = 0; }
    encode_init<AA>(enc_tables, threadIdx.x, blockDim.x);
    __syncthreads();

    // Flush: wave v owns the buckets [v * per_wave, (v + 1) * per_wave); one lane looks at one bucket; the wave then
    // writes its full buffers eight at a time, eight lanes per 128-byte group (16 bytes each: one coalesced line per
    // group instead of eight scattered 16-byte stores by one lane).  min_fill = kGroup: full buffers only;
    // min_fill = 1: every non-empty buffer, padded with fillers (end of the kernel).
    const uint32_t per_wave = (n_buckets + kScatterWaves - 1) / kScatterWaves;
    auto flush_wave = [&](uint32_t min_fill) {
        for (uint32_t b0 = (uint32_t)wave * per_wave; b0 < min((uint32_t)(wave + 1) * per_wave, n_buckets); b0 += 64) {
            const uint32_t b = b0 + (uint32_t)lane;
            const bool mine = b < min((uint32_t)(wave + 1) * per_wave, n_buckets);
            const uint32_t c = mine ? cnt[b] : 0u;
            unsigned long long dst_off = ~0ull;                       // entry index in ent (bit 62: in ovf_ent)
            if (c >= min_fill && c > 0) {
                if (c < kGroup)
                    for (uint32_t k = c; k < kGroup; k++) buf[(size_t)b * kGroup + k] = kEntInvalid;
                const uint32_t rel = wrel[b];
                if (rel + kGroup <= cap) {
                    dst_off = ((uint64_t)b * n_wg + w) * cap + rel;
                    wrel[b] = rel + kGroup;
                } else {
                    const uint32_t g = atomicAdd(ovf_cursor, 1u);
                    if (g < ovf_cap) { ovf_bucket[g] = b; dst_off = (1ull << 62) | ((uint64_t)g * kGroup); }
                    else dst_off = ~0ull - 1;                         // dropped (the host falls back to direct probing)
                }
                cnt[b] = 0;
            }
            unsigned long long m = __ballot(dst_off != ~0ull);
            wave_sync();                                             // the fillers above are read by other lanes below
            while (m) {
                // the next (up to) eight flushing lanes; lane group g = lane / 8 takes the g-th of them
                int src_lane = -1;
                unsigned long long mm = m;
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int ln = mm ? __builtin_ctzll(mm) : -1;
                    if (mm) mm &= mm - 1;
                    if ((lane >> 3) == g) src_lane = ln;
                }
                m = mm;
                const int sl = src_lane < 0 ? 0 : src_lane;
                const uint32_t fb = (uint32_t)__shfl((int)b, sl);
                const unsigned long long off = __shfl(dst_off, sl);
                if (src_lane >= 0 && off != ~0ull - 1) {
                    const uint32_t sub = (uint32_t)lane & 7u;
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(buf + (size_t)fb * kGroup + 2 * sub);
                    uint64_t *base = (off >> 62) & 1 ? ovf_ent + (off & ~(1ull << 62)) : ent + off;
                    store_entry_pair(base + 2 * sub, v);
                }
            }
        }
    };

    uint32_t n_valid = 0;                             // per lane: <= 6 x the wave's blocks
    bool ran_off = false;
    const uint32_t n_iter = (n_blocks + n_wg * kScatterWaves - 1) / (n_wg * kScatterWaves);
    // the next block's descriptor and characters are fetched while the current block is encoded
    BlockDesc bd_next;
    uint32_t raw_next[4] = {0, 0, 0, 0};
    {
        const uint32_t it0 = w * kScatterWaves + (uint32_t)wave;
        if (it0 < n_blocks) { bd_next = blocks[block_lo + it0]; load_block_chars<AA>(seq, bd_next, lane, raw_next); }
    }
    constexpr int RG = AA ? 1 : KG_SCATTER_RG;        // rows per group
    static_assert(ROWS % RG == 0, "KG_SCATTER_RG must divide 6");
    for (uint32_t iter = 0; iter < n_iter; iter++) {
        const uint32_t it = (iter * n_wg + w) * kScatterWaves + (uint32_t)wave;      // wave-uniform
        BlockDesc bd;
        const bool have = it < n_blocks;
        if (have) {
            bd = bd_next;
            uint32_t raw[4] = {raw_next[0], raw_next[1], raw_next[2], raw_next[3]};
            const uint32_t itn = it + n_wg * kScatterWaves;
            if (itn < n_blocks) { bd_next = blocks[block_lo + itn]; load_block_chars<AA>(seq, bd_next, lane, raw_next); }
            encode_chars<AA>(l, enc_tables, raw, lane);
        }
        bool lowc = false;                                                  // (wave-uniform) the block was set aside
#pragma unroll
        for (int g0 = 0; g0 < ROWS; g0 += RG) {
        uint64_t e[RG];
        uint32_t bk[RG];
        uint32_t pend = 0, vmask = 0;
        if (have && !lowc) {
#pragma unroll
            for (int k = 0; k < RG; k++) {
                const int r = g0 + k;
                uint32_t hi, lo, q;
                bool valid = row_halves<AA>(l, r, lane, bd, &hi, &lo);
                const uint32_t slot = split_fast(hi, lo, num_sigs, m35, &q);
                if (valid) vmask |= 1u << k;                            // query k-mers (KGJ:913-920), counted per block below
                if (valid && slot >= limit32) { ran_off = true; if (prog) progress_note_beyond(prog, slot); }   // (truncated table file)
                valid = valid && slot < limit32;                        // beyond the stream: never probed
                bk[k] = slot >> shift;
                const uint32_t low = (q << shift) | (slot & ((1u << shift) - 1u));
                const uint32_t id = window_key<AA>(bd, block_lo + it, r, lane);
                e[k] = ((uint64_t)id << 32) | low;
                if (valid) pend |= 1u << k;
            }
            // the wave's encode scratch is reused by its next block -- and, below, the first 64 bytes of it (base codes: not
            // read again once the codon codes are there) by the flush's rank -> lane map
            if (g0 + RG == ROWS) wave_sync();
            // Low-complexity sequence (homopolymers, short tandem repeats): most windows of the block are one k-mer or
            // two, and pushing them through one 16-entry buffer serialises the whole workgroup on it.  Such a block
            // is not inserted here: it goes on a list for lowc_blocks_kernel, which appends its entries in bulk.
            if (g0 == 0) {
                const unsigned long long m0 = __ballot((pend & 1u) != 0);
                if (__popcll(m0) >= 32) {
                    const uint32_t lead0 = (uint32_t)__builtin_amdgcn_readlane((int)bk[0], __builtin_ctzll(m0));
                    if (__popcll(__ballot((pend & 1u) && bk[0] == lead0)) >= 24) {
                        if (lane == 0) lowc_blocks[atomicAdd(lowc_cursor, 1u)] = block_lo + it;
                        vmask = 0;                                         // counted by lowc_blocks_kernel
                        pend = 0;
                        lowc = true;
                        if (RG != ROWS) wave_sync();                       // (the later row groups are skipped)
                    }
                }
            }
        }
        n_valid += (uint32_t)__popc(vmask);
        // Insert without workgroup barriers.  A bucket's buffer is a 16-entry group with two counters:
        //   cnt[b]      tickets: atomicAdd gives the entry's place; >= 16 means "full, try again"
        //   written[b]  stores issued; the lane whose increment makes it 16 owns the group: it (with seven helper
        //               lanes of its wave) copies the 128 bytes to the region, then reopens the buffer (written = 0,
        //               then cnt = 0).  Nobody else touches buf[b] / wrel[b] between the 16th ticket and the reopening.
        // LDS operations of one wave execute in program order and LDS is coherent in the workgroup, so the fences
        // below only pin the compiler's order.  A lane that keeps failing waits for a flush that the owning wave
        // performs right after its own stores; it polls the counter with plain reads in the meantime: retrying the
        // atomic instead (16 waves on one counter: homopolymer runs) starves the flushing wave and can run the
        // 32-bit ticket counter round to zero, which hands out a full buffer's slots again (seen as an intermittent
        // stall before the polling loop existed).  The guards turn any remaining protocol failure into the host's
        // fallback to the direct strategy (sticky word ovf_cursor[2], kg_stats.fallback == 2) instead of a hang or a
        // wrong result.
        uint32_t done = 0, spins = 0;
        if (insert_prio) set_wave_prio(insert_prio);                   // (uniform) few instructions between long LDS waits
        for (;;) {
            uint32_t at[RG];
#pragma unroll
            for (int r = 0; r < RG; r++)                             // the LDS atomics of all rows in flight together
                at[r] = (pend & (1u << r)) ? atomicAdd(&cnt[bk[r]], 1u) : kGroup;
            // (retry rounds look before they draw: a full buffer is polled with plain reads, so that waiting lanes
            //  neither serialise on the counter against the flushing wave nor run the counter round to zero)
#pragma unroll
            for (int r = 0; r < RG; r++)
                if (at[r] < kGroup) buf[(size_t)bk[r] * kGroup + at[r]] = e[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
            for (int r = 0; r < RG; r++)
                if (at[r] < kGroup) {
                    pend &= ~(1u << r);
                    if (atomicAdd(&written[bk[r]], 1u) == kGroup - 1) done |= 1u << r;
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // flush the groups completed by lanes of this wave: one per lane per pass, eight lanes per group
            for (;;) {
                const bool has = done != 0;
                unsigned long long m = __ballot(has);
                if (!m) break;
                const int r0 = has ? __builtin_ctz(done) : 0;
                uint32_t b = bk[0];
#pragma unroll
                for (int r = 1; r < RG; r++)
                    if (r0 == r) b = bk[r];
                unsigned long long dst_off = ~0ull;                    // entry index in ent (bit 62: in ovf_ent)
                bool to_ovf = false;
                if (has) {
                    done &= done - 1;
                    const uint32_t rel = wrel[b];
                    if (rel + kGroup <= cap) {
                        // (b * n_wg + w < 2^18 and cap < 2^24: 24-bit multiplies, full rate; the plain 64-bit expression
                        //  compiles to three quarter-rate v_mad_u64_u32 per flush pass)
                        dst_off = mul24_wide(b * n_wg + w, cap) + rel;
                        wrel[b] = rel + kGroup;
                    } else {
                        to_ovf = true;
                    }
                }
                const unsigned long long movf = __ballot(to_ovf);      // region full: overflow list, one atomic per pass
                if (movf) {
                    uint32_t og = 0;
                    if (lane == 0) og = atomicAdd(ovf_cursor, (uint32_t)__popcll(movf));
                    og = (uint32_t)__builtin_amdgcn_readfirstlane((int)og);
                    if (to_ovf) {
                        const uint32_t g = og + (uint32_t)__popcll(movf & ((1ull << lane) - 1ull));
                        if (g < ovf_cap) { ovf_bucket[g] = b; dst_off = (1ull << 62) | ((uint64_t)g * kGroup); }
                        else dst_off = ~0ull - 1;                     // dropped (the host falls back to direct probing)
                    }
                }
                // eight lanes copy one group; lane group g = lane / 8 serves the flushing lane of rank g + 8 * pass.
                // rank -> lane goes through 64 bytes of the wave's (idle) encode scratch instead of a scalar bit loop
                uint8_t *rank_lane = reinterpret_cast<uint8_t *>(&l);
                const uint32_t n_flush = (uint32_t)__popcll(m);
                if (has) rank_lane[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                wave_sync();
                for (uint32_t p0 = 0; p0 < n_flush; p0 += 8) {
                    const uint32_t want = p0 + ((uint32_t)lane >> 3);
                    const bool serve = want < n_flush;
                    const int sl = serve ? (int)rank_lane[want] : 0;
                    const uint32_t fb = (uint32_t)__shfl((int)b, sl);
                    const unsigned long long off = __shfl(dst_off, sl);
                    if (serve && off != ~0ull - 1) {
                        const uint32_t sub = (uint32_t)lane & 7u;
                        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(buf + (size_t)fb * kGroup + 2 * sub);
                        uint64_t *base = (off >> 62) & 1 ? ovf_ent + (off & ~(1ull << 62)) : ent + off;
                        store_entry_pair(base + 2 * sub, v);
                    }
                }
                wave_sync();                                           // the copies above read buf[b] before it reopens
                if (has) { written[b] = 0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); cnt[b] = 0; }
            }
            if (!__ballot(pend != 0)) break;
            // entries left: their buffers were full.  Wait until a lane's first waiting buffer reopens (bounded), then try
            // again.
            {
                const int rp = pend ? __builtin_ctz(pend) : 0;
                uint32_t pb = bk[0];
#pragma unroll
                for (int r = 1; r < RG; r++)
                    if (rp == r) pb = bk[r];
                const volatile uint32_t *pc = &cnt[pb];
                for (uint32_t polls = 0; polls < 64; polls++) {
                    if (__ballot(pend != 0 && *pc < kGroup)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // Never expected (see above).  The second test keeps a ticket counter far from wrapping round to zero,
            // which would hand out the slots of a full buffer a second time.
            bool runaway = false;
#pragma unroll
            for (int r = 0; r < RG; r++) runaway = runaway || (at[r] != kGroup && at[r] >= (1u << 28));
            if (++spins > (1u << 20) || __ballot(runaway)) {
                if (lane == 0) atomicOr(ovf_cursor + 2, 1u);           // sticky "protocol failure" word (ovf_cursor stays a group count)
                break;
            }
        }
        if (insert_prio) set_wave_prio(0);
        }   // row group
    }
    __syncthreads();
    // partial groups, padded with fillers
    flush_wave(1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x) fill[(uint64_t)b * n_wg + w] = wrel[b];
    for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off);
    if (lane == 0 && n_valid) atomicAdd(&ctr[0], (unsigned long long)n_valid);
    flush_ran_off(ran_off, ctr, lane);
}

// ---------------------------------------------------------------------------------------
// The blocks the scatter pass set aside (low-complexity sequence), one wave per block, after the scatter pass of the
// chunk: encode again, then every row's same-bucket sets of >= 16 entries are appended to a region of that bucket in
// one piece (padded to whole groups with fillers; regions are filled through fill[] with global atomics now -- the
// scatter workgroups have published them), or to the overflow list when the region is full; what is left goes in
// single entries.  Regions are picked by block number, so a long run spreads over all of a bucket's regions.
constexpr int kLowcWaves = 4;     // 3.4 KB of LDS per workgroup since the byte encode (0.7 KB per wave): the (usually idle) kernel
                                  // still finds room on a CU whose LDS a resident scatter workgroup has taken 105 KB of
template <bool AA>
__global__ __launch_bounds__(64 * kLowcWaves) void lowc_blocks_kernel(
    const uint8_t *__restrict__ seq, const BlockDesc *__restrict__ blocks, const uint32_t *__restrict__ lowc_cursor,
    const uint32_t *__restrict__ lowc_blocks, uint64_t limit, uint32_t num_sigs, uint32_t m35, uint32_t shift, uint32_t n_regions,
    uint32_t cap, uint64_t *__restrict__ ent, uint32_t *__restrict__ fill, uint32_t *ovf_cursor, uint32_t ovf_cap,
    uint32_t *__restrict__ ovf_bucket, uint64_t *__restrict__ ovf_ent, unsigned long long *ctr, Progress *prog)
{
    constexpr int ROWS = AA ? 1 : 6;
    __shared__ typename WaveLds<AA>::type lds[kLowcWaves];
    __shared__ typename WaveLds<AA>::tables enc_tables;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typename WaveLds<AA>::type &l = lds[wave];
    const uint32_t n_list = *lowc_cursor;
    if (n_list == 0) return;                                        // clean input: nothing was set aside (uniform)
    encode_init<AA>(enc_tables, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t limit32 = limit < 0xFFFFFFFFull ? (uint32_t)limit : 0xFFFFFFFFu;
    unsigned long long n_valid = 0;
    bool ran_off = false;
    // Overflow groups come from a wave-private pool reserved 32 at a time: one returning atomic on the list's cursor
    // per set would run at the ~90 per microsecond a single word sustains.  Unused pool groups are handed in empty.
    uint32_t pool_at = 0, pool_end = 0;                             // wave-uniform
    auto pool_flush = [&]() {
        for (uint32_t g = pool_at + ((uint32_t)lane >> 4); g < pool_end && g < ovf_cap; g += 4) {
            if ((lane & 15) == 0) ovf_bucket[g] = 0;
            ovf_ent[(uint64_t)g * kGroup + ((uint32_t)lane & 15u)] = kEntInvalid;
        }
        pool_at = pool_end;
    };
    // ng groups (16 entries each, fillers included) for bucket `lead`; nullptr when the list is full
    auto ovf_groups = [&](uint32_t lead, uint32_t ng) -> uint64_t * {
        if (pool_end - pool_at < ng) {
            pool_flush();
            const uint32_t want = ng > 32u ? ng : 32u;
            uint32_t g0 = 0;
            if (lane == 0) g0 = atomicAdd(ovf_cursor, want);
            g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0);
            pool_at = g0; pool_end = g0 + want;
        }
        const uint32_t g = pool_at;
        pool_at += ng;
        if (g + ng > ovf_cap) return nullptr;                       // dropped: the host falls back to direct probing
        if ((uint32_t)lane < ng) ovf_bucket[g + lane] = lead;
        return ovf_ent + (uint64_t)g * kGroup;
    };
    for (uint32_t i = blockIdx.x * kLowcWaves + (uint32_t)wave; i < n_list; i += gridDim.x * kLowcWaves) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)lowc_blocks[i]);
        const BlockDesc bd = blocks[it];
        encode_block<AA>(l, enc_tables, seq, bd, lane);
        const uint32_t region_w = it % n_regions;
        for (int r = 0; r < ROWS; r++) {
            uint32_t hi, lo, q;
            bool valid = row_halves<AA>(l, r, lane, bd, &hi, &lo);
            const uint32_t slot = split_fast(hi, lo, num_sigs, m35, &q);
            if (valid) n_valid++;
            if (valid && slot >= limit32) { ran_off = true; if (prog) progress_note_beyond(prog, slot); }
            bool pend = valid && slot < limit32;
            const uint32_t bkt = slot >> shift;
            const uint64_t e = ((uint64_t)window_key<AA>(bd, it, r, lane) << 32) | ((q << shift) | (slot & ((1u << shift) - 1u)));
            // big same-bucket sets first: one reservation and one coalesced store per set
            for (int pass = 0; pass < 4; pass++) {
                const unsigned long long mp = __ballot(pend);
                if (__popcll(mp) < 16) break;
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)bkt, __builtin_ctzll(mp));
                const bool mine = pend && bkt == lead;
                const unsigned long long same = __ballot(mine);
                const uint32_t n = (uint32_t)__popcll(same);
                if (n < 8) break;                                     // a diverse row: single entries below
                const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                const uint64_t region = (uint64_t)lead * n_regions + region_w;
                uint32_t rel = cap;                                   // a region known to be full is not touched again
                if (lane == 0 && *const_cast<volatile uint32_t *>(&fill[region]) < cap) rel = atomicAdd(&fill[region], n);
                rel = (uint32_t)__builtin_amdgcn_readfirstlane((int)rel);
                if (rel + n <= cap) {
                    if (mine) ent[region * cap + rel + rank] = e;
                } else {
                    // the set does not fit: what is left of the region stays empty (fillers), the set goes to the
                    // overflow list in whole groups
                    if (rel < cap && (uint32_t)lane < cap - rel) ent[region * cap + rel + lane] = kEntInvalid;
                    const uint32_t npad = (n + kGroup - 1) & ~(kGroup - 1);
                    uint64_t *dst = ovf_groups(lead, npad / kGroup);
                    if (dst) {
                        if (mine) dst[rank] = e;
                        if ((uint32_t)lane < npad - n) dst[n + lane] = kEntInvalid;
                    }
                }
                if (mine) pend = false;
            }
            // the rest one by one
            if (pend) {
                const uint64_t region = (uint64_t)bkt * n_regions + region_w;
                const uint32_t rel = atomicAdd(&fill[region], 1u);
                if (rel < cap) { ent[region * cap + rel] = e; pend = false; }
            }
            // ... and those whose region is full: overflow groups, one bucket per pass
            for (;;) {
                const unsigned long long mp = __ballot(pend);
                if (!mp) break;
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)bkt, __builtin_ctzll(mp));
                const bool mine = pend && bkt == lead;
                const unsigned long long same = __ballot(mine);
                const uint32_t n = (uint32_t)__popcll(same);
                const uint32_t npad = (n + kGroup - 1) & ~(kGroup - 1);
                const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                uint64_t *dst = ovf_groups(lead, npad / kGroup);
                if (dst) {
                    if (mine) dst[rank] = e;
                    if ((uint32_t)lane < npad - n) dst[n + lane] = kEntInvalid;
                }
                if (mine) pend = false;
            }
        }
        wave_sync();   // LDS is reused by the next block
    }
    pool_flush();
    for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off);
    if (lane == 0 && n_valid) atomicAdd(&ctr[0], n_valid);
    flush_ran_off(ran_off, ctr, lane);
}

// ---------------------------------------------------------------------------------------
// State of a wave's reservation in the unordered hit list.
struct UListState { unsigned long long base; uint32_t used; bool have; };

// Probe N entries of bucket b per lane and emit the hits: {key, payload} appended to the unordered list (wave-private
// chunk reservations).
template <bool AA, int N, bool COUNTERS>
__device__ __forceinline__ void probe_entries(const TableView &tab, uint32_t b, uint32_t shift, const uint64_t (&e)[N],
                                              kg_hit *__restrict__ ulist, uint32_t *__restrict__ chunk_used,
                                              unsigned long long *cursor, uint64_t ulist_cap, UListState &u,
                                              unsigned long long &ctr_slots, bool &ran_off, int lane, Progress *prog)
{
    uint64_t val[N], slot[N];
    bool valid[N];
    uint32_t id[N], fp[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        valid[k] = e[k] != kEntInvalid;
        const uint32_t low = (uint32_t)e[k];
        id[k] = (uint32_t)(e[k] >> 32);
        slot[k] = ((uint64_t)b << shift) | (low & ((1u << shift) - 1u));
        val[k] = (uint64_t)(low >> shift) * tab.num_sigs + slot[k];
        fp[k] = tag_qs(low >> shift, slot[k]);
    }
    Payload pay[N];
    unsigned long long ctr_dummy = 0;
    uint32_t fslot[N];
#pragma unroll
    for (int k = 0; k < N; k++) fslot[k] = 0;
    const uint32_t foundm = probe_n<N, COUNTERS>(tab, val, slot, fp, valid, pay, ctr_dummy, ctr_slots, ran_off, COUNTERS ? prog : nullptr,
                                                 COUNTERS ? fslot : nullptr);
    uint32_t cnt[N], rank[N], total = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const bool f = (foundm >> k) & 1u;
        const unsigned long long m = __ballot(f);
        cnt[k] = (uint32_t)__popcll(m);
        rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        total += cnt[k];
    }
    if (total) {
        if (u.used + total > kUChunk) {                      // uniform: retire the chunk, take a new one
            if (lane == 0 && u.have && u.base + kUChunk <= ulist_cap) chunk_used[u.base / kUChunk] = u.used;
            unsigned long long nb = 0;
            if (lane == 0) nb = atomicAdd(cursor, (unsigned long long)kUChunk);
            nb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) |
                 (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)nb);
            u.base = nb; u.used = 0; u.have = true;
        }
        if (u.base + kUChunk <= ulist_cap) {
            uint32_t at = u.used;
#pragma unroll
            for (int k = 0; k < N; k++) {
                if ((foundm >> k) & 1u) {
                    kg_hit h;
                    h.container = id[k];
                    h.from0InProt = (int32_t)fslot[k];       // (the slot it was found at, for KG_F_PROGRESS; the placement overwrites it)
                    h.oI = pay[k].oI; h.avgOffFromEnd = pay[k].avg; h.fI = pay[k].fI; h.functionWt = pay[k].wt;
                    ulist[u.base + at + rank[k]] = h;
                }
                at += cnt[k];
            }
        }
        u.used += total;
    }
}

// ---------------------------------------------------------------------------------------
// The bucket pass is split in two so that its inner loop only ever waits for the L2:
//   tag pass    : per entry, walk the tags (L2-resident) to the first empty slot or fingerprint match; a match
//                 becomes a 16-byte candidate record {value, id, slots walked}.  Never touches the 24-byte records.
//   verify pass : one lane per candidate: fetch the record (random line from HBM, thousands in flight), compare the
//                 key, emit the hit; a fingerprint collision keeps walking (generic, rare).
// home + quo * numSigs = the k-mer value (numSigs < 2^31 on this path; the multiplication is left to the verify pass:
// v_mad_u64_u32 runs at a quarter of the VALU rate and the tag pass would pay it in every batch that has a candidate)
struct CandRec { uint32_t home, quo, id, walked; };   // walked: slots from the home slot; kWalkOn: no record to check yet
constexpr uint32_t kWalkOn = 0x80000000u;
constexpr uint32_t kScanOn = 0x40000000u;   // the key is known to lie in the run that starts at home (home index): scan the records
constexpr uint32_t kWalkedMask = 0x3FFFFFFFu;
static_assert(sizeof(CandRec) == 16, "CandRec must be 16 bytes");

// reserve `total` (<= kUChunk) consecutive records of a chunked list for this wave; ~0 when the list is full
__device__ __forceinline__ unsigned long long chunk_reserve(UListState &u, uint32_t total, uint32_t *__restrict__ chunk_used,
                                                            unsigned long long *cursor, uint64_t cap, int lane)
{
    if (u.used + total > kUChunk) {                          // uniform: retire the chunk, take a new one
        if (lane == 0 && u.have && u.base + kUChunk <= cap) chunk_used[u.base / kUChunk] = u.used;
        unsigned long long nb = 0;
        if (lane == 0) nb = atomicAdd(cursor, (unsigned long long)kUChunk);
        nb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) |
             (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)nb);
        u.base = nb; u.used = 0; u.have = true;
    }
    const unsigned long long at = u.base + u.used;
    u.used += total;
    return u.base + kUChunk <= cap ? at : ~0ull;
}

__device__ __forceinline__ void chunk_finish(const UListState &u, uint32_t *__restrict__ chunk_used, uint64_t cap, int lane)
{
    if (lane == 0 && u.have && u.base + kUChunk <= cap) chunk_used[u.base / kUChunk] = u.used;
}

// Tag pass work distribution: the workgroups with blockIdx % 8 == x (one XCD under round-robin placement --
// measured, tools/xcd_affinity.hip; speed only, never correctness) draw tickets from ONE counter per group: ticket t
// is hand-out t % n_grabs of the group's bucket t / n_grabs, so the group walks the buckets b % 8 == x in order, one
// region at a time, and an XCD's L2 holds one bucket's tags (two at a hand-over).  (A counter per bucket cost every
// workgroup one more round trip per bucket -- the draw that finds the bucket exhausted: 0.15-0.25 ms of a 100-125 Mbp
// batch, nothing at 1 Gbp; profiles/r02_pipeline.md section 5.)
template <bool COUNTERS>
__global__ __launch_bounds__(256) KG_TAG_REGS void bucket_tag_kernel(
    const uint8_t *__restrict__ tags, uint64_t limit, uint64_t num_sigs, const uint64_t *__restrict__ ent,
    const uint32_t *__restrict__ fill, uint32_t n_regions, uint32_t cap, uint32_t n_buckets, uint32_t shift,
    uint32_t grab /* entry slots per hand-out, multiple of 256 * kProbeN */,
    uint32_t *next_region /* ticket counter of group x at [32 * x], zeroed */,
    CandRec *__restrict__ cand, uint32_t *__restrict__ cand_used, unsigned long long *cand_cursor, uint64_t cand_cap,
    unsigned long long *ctr, Progress *prog /* KG_F_PROGRESS (COUNTERS kernel), else null */)
{
    constexpr int N = kProbeN;
    const uint32_t lim32 = (uint32_t)limit;
    __shared__ uint32_t s_region;
    __shared__ __attribute__((aligned(16))) CandRec s_stage[4][kStageFlush + 64];      // per wave: < kStageFlush records waiting + <= 64 new ones
    CandRec *stage = s_stage[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    uint32_t sfill = 0;                                                    // (wave-uniform)
    const int lane = threadIdx.x & 63;
    unsigned long long ctr_slots = 0;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;      // "full": the first append takes a chunk

    // a grab = `grab` consecutive entry slots of one region (regions are cap slots long; slots past the region's
    // fill are empty grabs).  Measured: whole regions (256 per bucket) 10.8 ms, 2048-slot grabs 11.6 ms.
    const uint32_t kGrab = grab;
    const uint32_t grabs_per_region = (cap + kGrab - 1) / kGrab;
    const uint32_t n_grabs = n_regions * grabs_per_region;
    const uint32_t xg = blockIdx.x & 7u;
    const uint32_t n_tickets = xg < n_buckets ? ((n_buckets - xg + 7u) / 8u) * n_grabs : 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_region = atomicAdd(&next_region[xg * 32u], 1u);
        __syncthreads();
        const uint32_t tk = s_region;
        if (tk >= n_tickets) break;                 // the group's buckets are exhausted
        const uint32_t b = xg + 8u * (tk / n_grabs), g = tk % n_grabs;
        const uint32_t w = g / grabs_per_region, g0 = (g % grabs_per_region) * kGrab;
        // the region's fill is requested together with its first batch of entries (slots below cap are mapped; what lies
        // behind the fill is discarded below): one round trip less per hand-out
        const uint32_t fraw = fill[(uint64_t)b * n_regions + w];
        uint32_t n = cap;                               // until the fill has arrived
        const uint64_t *src = ent + ((uint64_t)b * n_regions + w) * cap;
        for (uint32_t c0 = g0; c0 < n && c0 < g0 + kGrab; c0 += 256u * N) {
            const uint32_t bound = n;
            // Per entry the lane keeps the entry's two words, its fingerprint and (after the compare) the slots walked; the
            // home slot and the quotient are recomputed from the low word where they are needed (two instructions) -- the
            // kernel has to stay within 64 VGPRs (kg_partition.hpp, "Register budgets").  Slots are below 2^31 on this
            // path (the scatter pass's split_fast needs numSigs < 2^31).
            uint32_t low[N], id[N], fp[N], skip[N], walked[N];
            uint32_t vmask = 0;
            Tags16 tg[N];
            const uint32_t smask = (1u << shift) - 1u, bbase = b << shift;
            // all entry loads first, then all tag loads: N independent L2 requests in flight per lane (loads
            // complete in order, so interleaving entry and tag loads serialises them)
            uint64_t ev[N];
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint32_t i = c0 + (uint32_t)k * 256u + threadIdx.x;
                ev[k] = i < bound ? __builtin_nontemporal_load(src + i) : kEntInvalid;
            }
            if (c0 == g0) {
                uint32_t f = fraw;
                asm volatile("; fill first used here" : "+s"(f));       // (keeps the scalar wait behind the entry loads)
                n = min(f, cap);                                        // (bulk appends may have run past the region)
            }
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint64_t e = c0 + (uint32_t)k * 256u + threadIdx.x < n ? ev[k] : kEntInvalid;
                if (e != kEntInvalid) vmask |= 1u << k;
                low[k] = (uint32_t)e;
                id[k] = (uint32_t)(e >> 32);
                const uint32_t home = bbase | (low[k] & smask);
                fp[k] = tag_qs(low[k] >> shift, home);
                const uint32_t cur = (uint32_t)probe_window(home, &skip[k]);     // home - skip (skip != 0: the window straddles a line)
                if ((vmask >> k) & 1u) tg[k] = load_tags(tags + cur);
            }
            // a window that holds neither an empty slot nor the fingerprint (2 % of the probes: straddling windows,
            // long clusters) is not walked here: it goes to the candidate list with kWalkOn set and the verify pass
            // continues the walk.  This keeps the hot loop free of the generic walk.
            uint32_t candm = 0, walkm = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                walked[k] = 0;
                if ((vmask >> k) & 1u) {
                    bool emp;
                    const int i = first_stop(tg[k], fp[k], &emp, skip[k]);
                    walked[k] = (uint32_t)i - skip[k];                          // slots from the home slot
                    if (i == 16) { candm |= 1u << k; walkm |= 1u << k; }
                    else if (!emp) candm |= 1u << k;
                    else {
                        const uint32_t home = bbase | (low[k] & smask), at = home + walked[k];
                        if (at >= lim32) ran_off = true;       // the "empty slot" is the padding behind the last record
                        if (COUNTERS) {
                            ctr_slots += (at < lim32 ? at + 1u : lim32) - home;
                            if (prog) progress_note_walk(prog, home, at, lim32);
                        }
                    }
                }
            }
            // (Resolving the undecided windows here with a second tag load -- 27 M of the 65 M candidates per Gbp, each of which
            //  costs the verify pass a random tag line -- was built and measured in round 3: verify 1.84 -> 1.55 ms alone, but
            //  this kernel, the stage's critical chain, 3.9 -> 4.1 ms per chunk beside the scatter pass: stage 20.3 -> 20.8 ms.
            //  profiles/r03_experiments.md.)
            // candidates -> the wave's staging buffer in LDS -> the list, 64 records (1 KB, eight whole lines) per store
            // instruction.  (Stored straight from the registers -- four store instructions per batch with ~3 active lanes
            // each -- the candidate output was 0.8 ms of the pass alone, for 1 GB of records.)
#pragma unroll
            for (int k = 0; k < N; k++) {
                const unsigned long long m = __ballot((candm >> k) & 1u);
                if (!m) continue;                                          // (uniform)
                if ((candm >> k) & 1u) {
                    CandRec c;
                    c.home = bbase | (low[k] & smask); c.quo = low[k] >> shift;
                    c.id = id[k]; c.walked = walked[k] | (((walkm >> k) & 1u) ? kWalkOn : 0u);
                    stage[sfill + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c;
                }
                sfill += (uint32_t)__popcll(m);                            // < kStageFlush + 64
                while (sfill >= kStageFlush) {
                    // (one wave's LDS operations execute in program order: the fences only pin the compiler's order)
                    wave_sync();
                    const unsigned long long at = chunk_reserve(u, kStageFlush, cand_used, cand_cursor, cand_cap, lane);
                    if ((uint32_t)lane < kStageFlush) {
                        const ulonglong2 out = *reinterpret_cast<const ulonglong2 *>(stage + lane);
                        if (at != ~0ull) stream_store16(cand + at + (uint32_t)lane, &out);
                    }
                    sfill -= kStageFlush;
                    wave_sync();
                    if ((uint32_t)lane < sfill) {                          // the remainder moves down
                        const ulonglong2 rest = *reinterpret_cast<const ulonglong2 *>(stage + kStageFlush + lane);
                        *reinterpret_cast<ulonglong2 *>(stage + lane) = rest;
                    }
                    wave_sync();
                }
            }
        }
    }
    if (sfill) {                                                           // the wave's last, partial group
        wave_sync();
        const CandRec out = stage[(uint32_t)lane < sfill ? lane : 0];
        const unsigned long long at = chunk_reserve(u, sfill, cand_used, cand_cursor, cand_cap, lane);
        if (at != ~0ull && (uint32_t)lane < sfill) stream_store16(cand + at + (uint32_t)lane, &out);
    }
    chunk_finish(u, cand_used, cand_cap, lane);
    flush_ran_off(ran_off, ctr, lane);
    if (COUNTERS) {
        for (int off = 32; off > 0; off >>= 1) ctr_slots += __shfl_down(ctr_slots, off);
        if (lane == 0) atomicAdd(&ctr[1], ctr_slots);
    }
}

// The same pass on the table's BYTE HOME INDEX (kg_device.hpp, build_bidx_kernel) instead of the tags: hand-outs, entry
// stream and candidate staging as in bucket_tag_kernel, but per entry ONE byte load out of the L2 (a bucket of the index is
// as large as a bucket of tags), one LDS read that decodes it (256-word table) and a bit test with the query's quotient:
//   listed, exact code      -> candidate flagged kScanOn: the key IS in the occupied run from its home slot; the verify pass
//                              scans the records there and takes the payload
//   listed, hashed / inexact -> candidate flagged kWalkOn: the verify pass walks the tags (generic)
//   not listed              -> a miss for certain; the reference's walk ends with the stream iff the home slot lies in the
//                              occupied run at its end (home >= tail_start: lookup_ran_off, KGJ:799-802)
// No fingerprint, no 16-tag window, no undecided window: ~40 VALU per entry against ~126, and 36.7 M + 6 M candidates per Gbp
// of the bench against 65 M.  Not used by KG_F_COUNTERS scans (slots_inspected needs the walk: bucket_tag_kernel<true>).
// N entries per lane and iteration, taken from R regions of the bucket at a time (R divides N): a hand-out costs the
// workgroup a ticket, two barriers and the HBM latency of its first entries -- ~5 us, whatever the region holds -- so the
// short regions of small chunks (a 125 Mbp shard in two chunks: 500 entries per region) are handed out two or four at a time and
// walked side by side, 256 * N / R slots of each per iteration: full lanes and half / a quarter of the hand-outs.
// EXACT: the table's quotients are all below 19, a class IS the quotient (no q % 19: two quarter-rate multiplies per entry,
// which the compiler otherwise computes for every entry and then selects away -- half of the pass's VALU time, r04 ISA).
template <int N, int R, bool EXACT>
__global__ __launch_bounds__(256) KG_TAG_REGS void bucket_index_kernel(
    const uint8_t *__restrict__ bidx, uint32_t tail_start,
    const uint64_t *__restrict__ ent, const uint32_t *__restrict__ fill, uint32_t n_regions /* multiple of R */, uint32_t cap, uint32_t n_buckets,
    uint32_t shift, uint32_t grab /* entry slots (of every region) per hand-out, multiple of 256 * N / R */,
    uint32_t *next_region /* ticket counter of group x at [32 * x], zeroed */,
    CandRec *__restrict__ cand, uint32_t *__restrict__ cand_used, unsigned long long *cand_cursor, uint64_t cand_cap,
    unsigned long long *ctr, uint32_t prio /* wave priority (0..3) */)
{
    static_assert(N % R == 0, "R must divide N");
    constexpr int PER = N / R;                          // entries per lane, region and iteration
    __shared__ uint32_t s_region;
    __shared__ uint32_t s_lut[256];
    __shared__ __attribute__((aligned(16))) CandRec s_stage[4][kStageFlush + 64];      // per wave: < kStageFlush records waiting + <= 64 new ones
    CandRec *stage = s_stage[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    uint32_t sfill = 0;                                                    // (wave-uniform)
    const int lane = threadIdx.x & 63;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;      // "full": the first append takes a chunk
    s_lut[threadIdx.x] = bidx_decode(threadIdx.x);      // (256 threads; the loop's first barrier publishes it)
    if (prio) set_wave_prio(prio);                      // a few instructions between L2 round trips: ahead of the scatter waves' VALU streams

    const uint32_t kGrab = grab;
    const uint32_t grabs_per_region = (cap + kGrab - 1) / kGrab;
    const uint32_t n_grabs = (n_regions / R) * grabs_per_region;
    const uint32_t xg = blockIdx.x & 7u;
    const uint32_t n_tickets = xg < n_buckets ? ((n_buckets - xg + 7u) / 8u) * n_grabs : 0u;
    constexpr uint32_t all_walk = EXACT ? 0u : kBidxInexact;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_region = atomicAdd(&next_region[xg * 32u], 1u);
        __syncthreads();
        const uint32_t tk = s_region;
        if (tk >= n_tickets) break;                 // the group's buckets are exhausted
        const uint32_t b = xg + 8u * (tk / n_grabs), g = tk % n_grabs;
        const uint32_t w = (g / grabs_per_region) * R, g0 = (g % grabs_per_region) * kGrab;
        // the regions' fills are requested together with their first batch of entries (slots below cap are mapped; what lies
        // behind the fill is discarded below): one round trip less per hand-out
        uint32_t fraw[R], n[R], nmax = cap;
#pragma unroll
        for (int r = 0; r < R; r++) { fraw[r] = fill[(uint64_t)b * n_regions + w + r]; n[r] = cap; }
        const uint64_t *src = ent + ((uint64_t)b * n_regions + w) * cap;          // region r of the hand-out: src + r * cap
        const uint32_t smask = (1u << shift) - 1u, bbase = b << shift;
        for (uint32_t c0 = g0; c0 < nmax && c0 < g0 + kGrab; c0 += 256u * PER) {
            uint64_t ev[N];
#pragma unroll
            for (int k = 0; k < N; k++) {
                const int r = k / PER;
                const uint32_t i = c0 + (uint32_t)(k % PER) * 256u + threadIdx.x;
                ev[k] = i < n[r] ? __builtin_nontemporal_load(src + (uint64_t)r * cap + i) : kEntInvalid;
            }
            if (c0 == g0) {
                nmax = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    uint32_t f = fraw[r];
                    asm volatile("; fill first used here" : "+s"(f));   // (keeps the scalar wait behind the entry loads)
                    n[r] = min(f, cap);                                 // (bulk appends may have run past the region)
                    nmax = max(nmax, n[r]);
                }
            }
            uint32_t code[N], vmask = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                if (c0 + (uint32_t)(k % PER) * 256u + threadIdx.x >= n[k / PER]) ev[k] = kEntInvalid;
                code[k] = 0;
                if (ev[k] != kEntInvalid) {
                    vmask |= 1u << k;
                    code[k] = bidx[bbase | ((uint32_t)ev[k] & smask)];      // the home slot's byte, out of the L2
                }
            }
            // branch-free: the N decodes' LDS reads are in flight together; an entry slot without an entry has code 0 = nothing listed
            uint32_t candm = 0, walkm = 0, wd[N];
#pragma unroll
            for (int k = 0; k < N; k++) wd[k] = s_lut[code[k]];
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint32_t q = (uint32_t)ev[k] >> shift;
                const uint32_t c = EXACT ? q : q % kBidxClasses;
                const uint32_t listed = (wd[k] >> c) & 1u;
                candm |= listed << k;
                walkm |= (listed & ((wd[k] | all_walk) >> 31)) << k;
                ran_off = ran_off || (((vmask >> k) & 1u) && !listed && (bbase | ((uint32_t)ev[k] & smask)) >= tail_start);
            }
#pragma unroll
            for (int k = 0; k < N; k++) {
                const unsigned long long m = __ballot((candm >> k) & 1u);
                if (!m) continue;                                          // (uniform)
                if ((candm >> k) & 1u) {
                    CandRec c;
                    c.home = bbase | ((uint32_t)ev[k] & smask); c.quo = (uint32_t)ev[k] >> shift;
                    c.id = (uint32_t)(ev[k] >> 32); c.walked = ((walkm >> k) & 1u) ? kWalkOn : kScanOn;
                    stage[sfill + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c;
                }
                sfill += (uint32_t)__popcll(m);                            // < kStageFlush + 64
                while (sfill >= kStageFlush) {
                    wave_sync();
                    const unsigned long long at = chunk_reserve(u, kStageFlush, cand_used, cand_cursor, cand_cap, lane);
                    if ((uint32_t)lane < kStageFlush) {
                        const ulonglong2 out = *reinterpret_cast<const ulonglong2 *>(stage + lane);
                        if (at != ~0ull) stream_store16(cand + at + (uint32_t)lane, &out);
                    }
                    sfill -= kStageFlush;
                    wave_sync();
                    if ((uint32_t)lane < sfill) {                          // the remainder moves down
                        const ulonglong2 rest = *reinterpret_cast<const ulonglong2 *>(stage + kStageFlush + lane);
                        *reinterpret_cast<ulonglong2 *>(stage + lane) = rest;
                    }
                    wave_sync();
                }
            }
        }
    }
    if (sfill) {                                                           // the wave's last, partial group
        wave_sync();
        const CandRec out = stage[(uint32_t)lane < sfill ? lane : 0];
        const unsigned long long at = chunk_reserve(u, sfill, cand_used, cand_cursor, cand_cap, lane);
        if (at != ~0ull && (uint32_t)lane < sfill) stream_store16(cand + at + (uint32_t)lane, &out);
    }
    chunk_finish(u, cand_used, cand_cap, lane);
    flush_ran_off(ran_off, ctr, lane);
}

// Verify pass: one wave per candidate chunk at a time, one lane per candidate.
template <bool AA, bool COUNTERS>
__global__ __launch_bounds__(256) void verify_kernel(
    const uint8_t *__restrict__ entries, const uint8_t *__restrict__ tags, uint64_t limit, uint64_t num_sigs, uint64_t magic,
    const CandRec *__restrict__ cand, const uint32_t *__restrict__ cand_used, const unsigned long long *cand_cursor,
    uint64_t cand_cap, kg_hit *__restrict__ ulist, uint32_t *__restrict__ chunk_used, unsigned long long *cursor,
    uint64_t ulist_cap, unsigned long long *ctr, Progress *prog /* KG_F_PROGRESS (COUNTERS kernels), else null */,
    uint32_t prio /* wave priority (0..3) */)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4;
    if (prio) set_wave_prio(prio);
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic; tab.m35 = 0;
    const unsigned long long cur = *cand_cursor;
    const uint32_t n_chunks = (uint32_t)((cur < cand_cap ? cur : cand_cap) / kUChunk);
    unsigned long long ctr_slots = 0;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;
    for (uint32_t c = wave_global; c < n_chunks; c += n_waves) {
        const uint32_t used = cand_used[c];
        for (uint32_t k0 = 0; k0 < used; k0 += 64) {
            const bool act = k0 + (uint32_t)lane < used;
            CandRec r;
            r.home = 0; r.quo = 0; r.id = 0; r.walked = 0;
            if (act) {
                const kg_u32x4 cv = __builtin_nontemporal_load(reinterpret_cast<const kg_u32x4 *>(cand + (uint64_t)c * kUChunk + k0 + lane));
                r.home = cv.x; r.quo = cv.y; r.id = cv.z; r.walked = cv.w;
            }
            const uint64_t quo = r.quo, home = r.home;
            const uint64_t val = quo * num_sigs + home;
            uint64_t s = home + (r.walked & kWalkedMask);
            bool found = false;
            Entry e;
            e.key = 0; e.oI = e.avg = e.fI = 0; e.wt = 0.f;
            if (act && (r.walked & kScanOn)) {
                // listed in the home index: the key is in the occupied run from its home slot; the first record that
                // carries it is the one the reference finds (KGJ:1003-1015)
                // (three keys are requested together: a key sits 0.5 slots behind its home slot on average at load 0.5, and
                //  the slowest lane of the wave sets the pace; records of neighbouring slots share their 128-byte line)
#if KG_SCAN_FULL
                // the record AT the home slot whole (two out of three listed keys sit there: a key is 0.5 slots behind its
                // home slot on average at load 0.5) and the keys of the two records behind it, all in flight together: one
                // round trip instead of two (keys, then the payload) for most candidates
                if (s < limit) {
                    const uint64_t s1 = s + 1 < limit ? s + 1 : s, s2 = s + 2 < limit ? s + 2 : s;
                    const Entry e0 = load_entry(tab, s);
                    const uint2 k1 = *reinterpret_cast<const uint2 *>(tab.entries + s1 * 24);
                    const uint2 k2 = *reinterpret_cast<const uint2 *>(tab.entries + s2 * 24);
                    const uint32_t vlo = (uint32_t)val, vhi = (uint32_t)(val >> 32);
                    if (e0.key == (int64_t)val) { found = true; e = e0; }
                    else if (s + 1 < limit && k1.x == vlo && k1.y == vhi) { found = true; s += 1; e = load_entry(tab, s); }
                    else if (s + 2 < limit && k2.x == vlo && k2.y == vhi) { found = true; s += 2; e = load_entry(tab, s); }
                    else s += 3;
                }
#endif
                while (!found) {
                    if (s >= limit) { s = limit; ran_off = true; break; }      // (only if the table changed under the index)
                    const uint64_t s1 = s + 1 < limit ? s + 1 : s, s2 = s + 2 < limit ? s + 2 : s;
                    const uint2 k0 = *reinterpret_cast<const uint2 *>(tab.entries + s * 24);
                    const uint2 k1 = *reinterpret_cast<const uint2 *>(tab.entries + s1 * 24);
                    const uint2 k2 = *reinterpret_cast<const uint2 *>(tab.entries + s2 * 24);
                    const uint32_t vlo = (uint32_t)val, vhi = (uint32_t)(val >> 32);
                    if (k0.x == vlo && k0.y == vhi) found = true;
                    else if (s + 1 < limit && k1.x == vlo && k1.y == vhi) { found = true; s += 1; }
                    else if (s + 2 < limit && k2.x == vlo && k2.y == vhi) { found = true; s += 2; }
                    if (found) { e = load_entry(tab, s); break; }
                    s += 3;
                }
                if (COUNTERS) {
                    ctr_slots += (s < limit ? s + 1 : limit) - home;
                    if (prog) progress_note_walk(prog, home, s, limit);
                }
            } else if (act) {
                if (!(r.walked & kWalkOn)) {               // a fingerprint match at s: check the record
                    e = load_entry(tab, s);
                    found = e.key == (int64_t)val;
                    if (!found) s += 1;                    // fingerprint collision: keep walking (KGJ:944-1034 semantics)
                }
                if (!found) {
                    const uint32_t f = tag_qs(quo, home);
                    for (;;) {
                        if (s >= limit) { s = limit; ran_off = true; break; }
                        const Tags16 x = load_tags(tab.tags + s);
                        bool emp;
                        const int i = first_stop(x, f, &emp);
                        if (i == 16) { s += 16; continue; }
                        s += (uint64_t)i;
                        if (emp) { if (s >= limit) ran_off = true; break; }
                        e = load_entry(tab, s);
                        if (e.key == (int64_t)val) { found = true; break; }
                        s += 1;
                    }
                }
                if (COUNTERS) {
                    ctr_slots += (s < limit ? s + 1 : limit) - home;
                    if (prog) progress_note_walk(prog, home, s, limit);
                }
            }
            const unsigned long long m = __ballot(found);
            const uint32_t total = (uint32_t)__popcll(m);
            if (total) {
                const unsigned long long at = chunk_reserve(u, total, chunk_used, cursor, ulist_cap, lane);
                if (at != ~0ull && found) {
                    kg_hit h;
                    h.container = r.id;
                    h.from0InProt = (int32_t)(uint32_t)s;    // (the slot it was found at, for KG_F_PROGRESS; the placement overwrites it)
                    h.oI = e.oI; h.avgOffFromEnd = e.avg; h.fI = e.fI; h.functionWt = e.wt;
                    stream_store_hit(ulist + at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), h);
                }
            }
        }
    }
    chunk_finish(u, chunk_used, ulist_cap, lane);
    flush_ran_off(ran_off, ctr, lane);
    if (COUNTERS) {
        for (int off = 32; off > 0; off >>= 1) ctr_slots += __shfl_down(ctr_slots, off);
        if (lane == 0) atomicAdd(&ctr[1], ctr_slots);
    }
}

// Overflow groups (regions that filled up: heavily repeated k-mers): one wave per 16-entry group.
template <bool AA, bool COUNTERS>
__global__ __launch_bounds__(256) void overflow_probe_kernel(
    const uint8_t *__restrict__ entries, const uint8_t *__restrict__ tags, uint64_t limit, uint64_t num_sigs, uint64_t magic,
    const uint32_t *__restrict__ ovf_bucket, const uint64_t *__restrict__ ovf_ent, const uint32_t *__restrict__ ovf_cursor,
    uint32_t ovf_cap, uint32_t shift, kg_hit *__restrict__ ulist, uint32_t *__restrict__ chunk_used, unsigned long long *cursor, uint64_t ulist_cap,
    unsigned long long *ctr, Progress *prog)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4;
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic; tab.m35 = 0;
    unsigned long long ctr_slots = 0;
    bool ran_off = false;
    UListState u;
    u.base = 0; u.used = kUChunk; u.have = false;
    const uint32_t n_groups = min(*ovf_cursor, ovf_cap);
    for (uint32_t g = wave_global; g < n_groups; g += n_waves) {
        const uint32_t b = ovf_bucket[g];
        uint64_t e[1];
        e[0] = lane < (int)kGroup ? ovf_ent[(uint64_t)g * kGroup + lane] : kEntInvalid;
        probe_entries<AA, 1, COUNTERS>(tab, b, shift, e, ulist, chunk_used, cursor, ulist_cap, u, ctr_slots, ran_off, lane, prog);
    }
    if (lane == 0 && u.have && u.base + kUChunk <= ulist_cap) chunk_used[u.base / kUChunk] = u.used;
    flush_ran_off(ran_off, ctr, lane);
    if (COUNTERS) {
        for (int off = 32; off > 0; off >>= 1) ctr_slots += __shfl_down(ctr_slots, off);
        if (lane == 0) atomicAdd(&ctr[1], ctr_slots);
    }
}

// ---------------------------------------------------------------------------------------
// A chunk of the batch starts and ends at sequence boundaries, so its rows are a contiguous range of the container-major
// row order and its hits a contiguous range of hits[] that starts at *base (the hits of the chunks before it; device
// memory, chained here).  This lets chunk c be ordered (kg_order.hpp) while chunk c+1 is still being scattered and probed.
// base[1] = base[0] + *chunk_total; the last chunk also publishes the grand total
__global__ void chunk_base_kernel(const uint64_t *chunk_total, uint64_t *base, uint64_t *grand_total)
{
    const uint64_t b = base[0] + *chunk_total;
    base[1] = b;
    if (grand_total) *grand_total = b;
}

}  // namespace kg
