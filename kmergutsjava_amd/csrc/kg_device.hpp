// kg_device.hpp -- gfx950 (CDNA4, wave64) kernels of the kmer_guts hot path.
//
// "KGJ:n" = reference lib/src/kmergutsjava/KmerGutsJava.java line n.  Nothing here is a
// translation of the reference's control flow: the reference materialises every query k-mer,
// sorts them by hash slot and merge-joins them with a byte stream of the table
// (KGJ:900-922, 1076-1095, 944-1034); here one wavefront owns one "window block" of a
// sequence, encodes its windows out of LDS, probes a tag array in HBM directly and compacts
// the hits with wave ballots.  The results are defined by the reference lines cited at each
// step and must be bit-identical.
//
// Work decomposition
//   DNA : block = 192 consecutive forward base positions p of one contig.  Position p starts
//         one '+' window (frame p%3, residue p/3) and one '-' window (the reverse-complement
//         window covering the same 24 bases), so a block carries 6 rows of 64 windows:
//         rows 0..2 = '+' strand phase 0..2, rows 3..5 = '-' strand phase 0..2, lane t of row
//         phase f handles p = 192*j + 3*t + f.  Each row belongs to exactly one of the six
//         HitContainers of the contig (KGJ:1064-1072) and is contiguous in residue index.
//   AA  : block = 64 consecutive windows of one protein, one row.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kmerguts_hip.h"

namespace kg {

constexpr int kWave = 64;
constexpr int kWavesPerWG = 4;
constexpr int kDnaPosPerBlock = 192;
constexpr int kAaWinPerBlock = 64;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kTagEmpty = 0xFFu;
constexpr uint64_t kNotFound = ~0ull;
constexpr int kTagPad = 64;            // EMPTY tags appended behind the last slot

// One work item ("window block"), 32 bytes, read with one scalar load.
struct BlockDesc {
    uint64_t soff;    // byte offset of the sequence inside the batch
    uint32_t len;     // sequence length in characters
    uint32_t j;       // block index inside the sequence
    uint32_t nk;      // number of blocks of this sequence
    uint32_t ibase;   // index of the sequence's first block
    uint32_t seq;     // sequence index in the batch
    uint32_t pad;
};
static_assert(sizeof(BlockDesc) == 32, "BlockDesc must be 32 bytes");

// Device view of the signature table.
//   entries : the 24-byte records of kmer.table.mem_map as they are on disk (KGJ:995-999)
//   tags    : one byte per slot: 0xFF = empty slot (whichKmer > 20^8, KGJ:1000), otherwise an
//             8-bit fingerprint of whichKmer in 0..0xFE.  A probe walks tags (16 slots per
//             load) and touches the 24-byte record only on a fingerprint match.
struct TableView {
    const uint8_t *entries;
    const uint8_t *tags;
    uint64_t limit;      // complete records present; the reference's stream ends here (EOF == not found)
    uint64_t num_sigs;   // modulus of the home slot (KGJ:969)
    uint64_t magic;      // floor(2^64 / num_sigs)
};

struct ScanArgs {
    TableView tab;
    const uint8_t *seq;
    const BlockDesc *blocks;
    uint32_t n_blocks;
    uint32_t *counts;            // hits per (virtual row); DNA 6 per block, AA 1 per block
    uint32_t *block_stage_base;  // first staging record of the block
    kg_hit *stage;               // staging area, block-granular placement by atomic cursor
    unsigned long long *cursor;  // staging cursor
    uint64_t stage_cap;
    unsigned long long *ctr;     // [0] windows_valid, [1] slots_inspected (KG_F_COUNTERS)
};

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
    // LDS traffic of one wave is executed in issue order; this only stops the compiler from
    // moving LDS reads of other lanes' data above the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t tag_of(uint64_t k)
{
    uint32_t h = (uint32_t)k * 0x9E3779B1u ^ (uint32_t)(k >> 32) * 0x85EBCA6Bu;
    h ^= h >> 15;
    uint32_t t = h >> 24;
    return t == kTagEmpty ? 0xFEu : t;
}

// exact v % num_sigs for v < 2^63 (one correction step suffices: q_est in {q-1, q})
__device__ __forceinline__ uint64_t home_slot(uint64_t v, const TableView &t)
{
    uint64_t q = __umul64hi(v, t.magic);
    uint64_t r = v - q * t.num_sigs;
    if (r >= t.num_sigs) r -= t.num_sigs;
    return r;
}

// 0x80 in every byte of x that is zero, nothing else (no cross-byte carries)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

struct Tags16 { uint32_t w[4]; };

__device__ __forceinline__ Tags16 load_tags(const uint8_t *p)
{
    Tags16 r;
    __builtin_memcpy(&r, p, 16);     // one global_load_dwordx4 at byte alignment
    return r;
}

// First slot i in 0..15 whose tag is EMPTY or == fp.  Returns 16 if none.  *is_empty tells which.
__device__ __forceinline__ int first_stop(const Tags16 &x, uint32_t fp, bool *is_empty)
{
    const uint32_t fpw = fp * 0x01010101u;
    uint32_t e0 = zero_bytes(~x.w[0]), e1 = zero_bytes(~x.w[1]), e2 = zero_bytes(~x.w[2]), e3 = zero_bytes(~x.w[3]);
    uint32_t f0 = zero_bytes(x.w[0] ^ fpw), f1 = zero_bytes(x.w[1] ^ fpw), f2 = zero_bytes(x.w[2] ^ fpw), f3 = zero_bytes(x.w[3] ^ fpw);
    uint64_t elo = ((uint64_t)e1 << 32) | e0, ehi = ((uint64_t)e3 << 32) | e2;
    uint64_t slo = elo | (((uint64_t)f1 << 32) | f0), shi = ehi | (((uint64_t)f3 << 32) | f2);
    if (slo) {
        int b = __builtin_ctzll(slo);
        *is_empty = (elo >> b) & 1;
        return b >> 3;
    }
    if (shi) {
        int b = __builtin_ctzll(shi);
        *is_empty = (ehi >> b) & 1;
        return 8 + (b >> 3);
    }
    *is_empty = false;
    return 16;
}

struct Entry { int64_t key; int32_t oI, avg, fI; float wt; };

__device__ __forceinline__ Entry load_entry(const TableView &t, uint64_t slot)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(t.entries + slot * 24);   // 8-byte aligned
    uint2 a = p[0], b = p[1], c = p[2];
    Entry e;
    e.key = (int64_t)(((uint64_t)a.y << 32) | a.x);
    e.oI = (int32_t)b.x; e.avg = (int32_t)b.y; e.fI = (int32_t)c.x; e.wt = __uint_as_float(c.y);
    return e;
}

// Generic probe from slot s (KGJ:944-1034 semantics: walk forward until the k-mer, an empty
// slot or the end of the stream; never wrap).  Returns the matching slot or kNotFound and
// the slot at which the walk stopped (for the inspected-entries counter).
__device__ __noinline__ uint64_t probe_slow(const TableView &t, uint64_t v, uint32_t fp, uint64_t s,
                                            Entry *hit, uint64_t *stop_slot)
{
    for (;;) {
        if (s >= t.limit) { *stop_slot = t.limit; return kNotFound; }
        Tags16 x = load_tags(t.tags + s);
        bool emp;
        int i = first_stop(x, fp, &emp);
        if (i == 16) { s += 16; continue; }
        if (emp) { *stop_slot = s + i; return kNotFound; }
        Entry e = load_entry(t, s + i);
        if (e.key == (int64_t)v) { *hit = e; *stop_slot = s + i; return s + i; }
        s += i + 1;
    }
}

// ---------------------------------------------------------------------------------------
// genetic code (KGJ:88-93) folded with toAminoAcidOff (KGJ:111-175): codon -> 0..19, stop -> 20.
// kCodon16[c1*4+c2] packs the four c3 codes, 5 bits each.
constexpr char kGeneticCode[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
constexpr uint32_t aa_code_of(char c)
{
    const char *alpha = "ACDEFGHIKLMNPQRSTVWY";
    for (uint32_t i = 0; i < 20; i++)
        if (alpha[i] == c) return i;
    return 20;
}
constexpr uint32_t codon16(int i)
{
    return aa_code_of(kGeneticCode[i * 4]) | (aa_code_of(kGeneticCode[i * 4 + 1]) << 5) |
           (aa_code_of(kGeneticCode[i * 4 + 2]) << 10) | (aa_code_of(kGeneticCode[i * 4 + 3]) << 15);
}
__constant__ uint32_t kCodon16[16] = {
    codon16(0), codon16(1), codon16(2), codon16(3), codon16(4), codon16(5), codon16(6), codon16(7),
    codon16(8), codon16(9), codon16(10), codon16(11), codon16(12), codon16(13), codon16(14), codon16(15)};

// dnaChar (KGJ:294-318): a/A 0, c/C 1, g/G 2, t/T/u/U 3, anything else 4
__device__ __forceinline__ uint32_t dna_code(uint32_t c)
{
    uint32_t u = c & 0xDFu;   // clears only bit 5: 'a'..'z' -> 'A'..'Z', nothing else becomes a letter
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

struct __attribute__((aligned(16))) WaveLdsDna {
    uint32_t H[208];      // '+' half codes: 4 codons starting at base q   (q < 204)
    uint32_t G[208];      // '-' half codes: 4 reverse-complement codons over bases q..q+11
    uint32_t t16[16];
    uint8_t bc[232];      // base codes of the block's 215 bases
    uint8_t F[224];       // aa code of forward codon starting at base q   (q < 213)
    uint8_t R[224];       // aa code of reverse-complement codon over bases q..q+2
};

struct __attribute__((aligned(16))) WaveLdsAa {
    uint32_t H4[80];      // half codes of 4 residues starting at q (q < 68)
    uint8_t code[80];
    uint8_t lut[256];
};

// ---------------------------------------------------------------------------------------
// The scan kernel.  ROWS = 6 (DNA) or 1 (AA).
template <bool AA, bool COUNTERS>
__global__ __launch_bounds__(kWave *kWavesPerWG) void scan_kernel(ScanArgs a)
{
    constexpr int ROWS = AA ? 1 : 6;
    __shared__ WaveLdsDna lds_dna[AA ? 1 : kWavesPerWG];
    __shared__ WaveLdsAa lds_aa[AA ? kWavesPerWG : 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerWG + wave);
    const uint32_t n_waves = gridDim.x * kWavesPerWG;
    const TableView tab = a.tab;

    WaveLdsDna &ld = lds_dna[AA ? 0 : wave];
    WaveLdsAa &la = lds_aa[AA ? wave : 0];
    if (AA) {
        // toAminoAcidOff (KGJ:111-175) as a 256-entry table: uppercase letters only
        for (int b = lane; b < 256; b += 64) la.lut[b] = (uint8_t)aa_code_of((char)b);
    } else {
        if (lane < 16) ld.t16[lane] = kCodon16[lane];
    }
    wave_sync();

    unsigned long long ctr_valid = 0, ctr_slots = 0;

    for (uint32_t it = wave_global; it < a.n_blocks; it += n_waves) {
        const BlockDesc bd = a.blocks[it];
        const uint64_t soff = bd.soff;
        const uint32_t L = bd.len, j = bd.j, nk = bd.nk;

        uint64_t val[ROWS];      // encodedKmer (KGJ:274-292)
        bool valid[ROWS];
        int32_t pos[ROWS];       // from0InProt
        uint32_t vrow[ROWS];     // index into counts[]
        uint32_t cont[ROWS];     // HitContainer id (KGJ:907-911 order)

        if (AA) {
            // ---- protein: windows i = 64j + lane, queried iff i < len - 8 (KGJ:912: i < pIseq.length - K)
            const uint32_t w0 = j * kAaWinPerBlock;
            const uint32_t nload = min(71u, L - w0);
            for (uint32_t q = lane; q < 72; q += 64) {
                uint32_t c = q < nload ? a.seq[soff + w0 + q] : 0u;
                la.code[q] = la.lut[c];          // own lane's write is read back by the same lane
            }
            wave_sync();
            for (uint32_t q = lane; q < 68; q += 64) {
                uint32_t c0 = la.code[q], c1 = la.code[q + 1], c2 = la.code[q + 2], c3 = la.code[q + 3];
                bool ok = (c0 < 20) & (c1 < 20) & (c2 < 20) & (c3 < 20);
                la.H4[q] = ok ? c0 * 8000u + c1 * 400u + c2 * 20u + c3 : kInvalid;
            }
            wave_sync();
            uint32_t hi = la.H4[lane], lo = la.H4[lane + 4];
            uint32_t i = w0 + lane;
            valid[0] = (hi != kInvalid) & (lo != kInvalid) & ((uint64_t)i + 8 < (uint64_t)L);
            val[0] = (uint64_t)hi * 160000ull + lo;
            pos[0] = (int32_t)i;
            vrow[0] = it;
            cont[0] = bd.seq;
            wave_sync();   // LDS is reused by the next block
        } else {
            // ---- DNA: stage 215 bases, derive codon codes for both strands, then 4-codon half codes
            const uint32_t ts = j * kDnaPosPerBlock;
            const uint32_t nload = min(215u, L - ts);
            for (uint32_t q = lane; q < 232; q += 64) {
                uint32_t c = q < nload ? a.seq[soff + ts + q] : (uint32_t)'N';
                ld.bc[q] = (uint8_t)dna_code(c);
            }
            wave_sync();
            for (uint32_t q = lane; q < 213; q += 64) {
                uint32_t b0 = ld.bc[q], b1 = ld.bc[q + 1], b2 = ld.bc[q + 2];
                bool ok = (b0 < 4) & (b1 < 4) & (b2 < 4);
                // translate (KGJ:320-343): codon index c1*16+c2*4+c3; non-ACGTU -> 'x' -> code 20
                uint32_t f = (ld.t16[(b0 * 4 + b1) & 15] >> ((b2 & 3) * 5)) & 31u;
                // reverse strand (KGJ:263-272 + 320-343): codon = compl(b2) compl(b1) compl(b0); compl code = 3 - code
                uint32_t r = (ld.t16[((3 - b2) * 4 + (3 - b1)) & 15] >> (((3 - b0) & 3) * 5)) & 31u;
                ld.F[q] = (uint8_t)(ok ? f : 20u);
                ld.R[q] = (uint8_t)(ok ? r : 20u);
            }
            wave_sync();
            for (uint32_t q = lane; q < 204; q += 64) {
                uint32_t f0 = ld.F[q], f1 = ld.F[q + 3], f2 = ld.F[q + 6], f3 = ld.F[q + 9];
                uint32_t r0 = ld.R[q], r1 = ld.R[q + 3], r2 = ld.R[q + 6], r3 = ld.R[q + 9];
                bool okf = (f0 < 20) & (f1 < 20) & (f2 < 20) & (f3 < 20);
                bool okr = (r0 < 20) & (r1 < 20) & (r2 < 20) & (r3 < 20);
                ld.H[q] = okf ? f0 * 8000u + f1 * 400u + f2 * 20u + f3 : kInvalid;
                // on the '-' strand the codon over the highest bases comes first
                ld.G[q] = okr ? r3 * 8000u + r2 * 400u + r1 * 20u + r0 : kInvalid;
            }
            wave_sync();
            const uint32_t vbase = 6u * bd.ibase;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const uint32_t pl = 3u * lane + f;
                uint32_t h0 = ld.H[pl], h1 = ld.H[pl + 12];
                uint32_t g0 = ld.G[pl], g1 = ld.G[pl + 12];
                // '+' strand: frame f (block start is a multiple of 3), residue index 64j + lane
                valid[f] = (h0 != kInvalid) & (h1 != kInvalid);
                val[f] = (uint64_t)h0 * 160000ull + h1;
                pos[f] = (int32_t)(j * 64u + lane);
                vrow[f] = vbase + (uint32_t)f * nk + j;
                cont[f] = bd.seq * 6u + (uint32_t)f;
                // '-' strand: the window over forward bases p..p+23 starts at reverse-complement base
                // b' = L-24-p, i.e. frame b'%3, residue b'/3 (KGJ:1068-1072).  24 % 3 == 0 and ts % 3 == 0,
                // so the frame depends on f only.
                const uint32_t fr = (L - (uint32_t)f) % 3u;
                valid[3 + f] = (g0 != kInvalid) & (g1 != kInvalid);
                val[3 + f] = (uint64_t)g1 * 160000ull + g0;
                pos[3 + f] = (int32_t)((L - 24u - (uint32_t)f - ts) / 3u) - (int32_t)lane;
                vrow[3 + f] = vbase + (3u + fr) * nk + (nk - 1u - j);
                cont[3 + f] = bd.seq * 6u + 3u + fr;
            }
            wave_sync();   // LDS is reused by the next block
        }

        // ---- probe: home slot, 16 tags per load, records touched only on a fingerprint match
        uint64_t slot[ROWS];
        uint32_t fp[ROWS];
        Tags16 tg[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            slot[r] = home_slot(val[r], tab);
            fp[r] = tag_of(val[r]);
            if (COUNTERS && valid[r]) ctr_valid++;          // query k-mers (KGJ:913-920)
            valid[r] = valid[r] && slot[r] < tab.limit;     // beyond the stream: EOF, not found, nothing inspected
            if (valid[r]) tg[r] = load_tags(tab.tags + slot[r]);
        }
        // state per row: 0 = resolved, 1 = candidate at cand[r], 2 = continue with the generic walk
        int st[ROWS];
        uint64_t cand[ROWS];
        uint64_t stop[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            st[r] = 0; cand[r] = kNotFound; stop[r] = slot[r];
            if (valid[r]) {
                bool emp;
                int i = first_stop(tg[r], fp[r], &emp);
                if (i == 16) { st[r] = 2; cand[r] = slot[r] + 16; }
                else if (emp) { stop[r] = slot[r] + (uint64_t)i; }
                else { st[r] = 1; cand[r] = slot[r] + (uint64_t)i; }
            }
        }
        Entry ent[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            if (st[r] == 1) ent[r] = load_entry(tab, cand[r]);
        bool found[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            found[r] = false;
            if (st[r] == 1) {
                if (ent[r].key == (int64_t)val[r]) { found[r] = true; st[r] = 0; stop[r] = cand[r]; }
                else { st[r] = 2; cand[r] = cand[r] + 1; }       // fingerprint collision: keep walking
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (st[r] == 2) {
                uint64_t s = probe_slow(tab, val[r], fp[r], cand[r], &ent[r], &stop[r]);
                found[r] = s != kNotFound;
                if (found[r]) cand[r] = s;
            }
        }
        if (COUNTERS) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (valid[r]) {
                    uint64_t last = stop[r] < tab.limit ? stop[r] + 1 : tab.limit;
                    ctr_slots += last - slot[r];
                }
            }
        } else {
            (void)stop;
        }

        // ---- ordered compaction: ballot per row, one staging reservation per wave
        uint32_t cnt[ROWS], rank[ROWS];
        uint32_t total = 0;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            unsigned long long m = __ballot(found[r]);
            cnt[r] = (uint32_t)__popcll(m);
            // '+' rows ascend with the lane, '-' rows descend: rank so that staging order == position order
            unsigned long long below = m & ((1ull << lane) - 1ull);
            unsigned long long above = lane == 63 ? 0ull : (m >> (lane + 1));
            rank[r] = (!AA && r >= 3) ? (uint32_t)__popcll(above) : (uint32_t)__popcll(below);
            total += cnt[r];
        }
        unsigned long long base = 0;
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) a.counts[vrow[r]] = cnt[r];
            if (total) base = atomicAdd(a.cursor, (unsigned long long)total);
            a.block_stage_base[it] = (uint32_t)base;
        }
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)base);
        if (total && base + total <= a.stage_cap) {
            uint32_t rowbase = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (found[r]) {
                    kg_hit h;
                    h.container = cont[r];
                    h.from0InProt = pos[r];
                    h.oI = ent[r].oI; h.avgOffFromEnd = ent[r].avg; h.fI = ent[r].fI; h.functionWt = ent[r].wt;
                    a.stage[base + rowbase + rank[r]] = h;
                }
                rowbase += cnt[r];
            }
        }
    }

    if (COUNTERS) {
        // wave reduction, one atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            ctr_valid += __shfl_down(ctr_valid, off);
            ctr_slots += __shfl_down(ctr_slots, off);
        }
        if (lane == 0) {
            atomicAdd(&a.ctr[0], ctr_valid);
            atomicAdd(&a.ctr[1], ctr_slots);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Block descriptors: block i -> sequence by binary search in the per-sequence prefix.
__global__ void build_blocks_kernel(const int64_t *seq_off, const uint32_t *ibase, uint32_t n_seqs,
                                    uint32_t n_blocks, BlockDesc *out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    uint32_t lo = 0, hi = n_seqs;       // largest k with ibase[k] <= i  (ibase[n_seqs] == n_blocks > i)
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ibase[mid] <= i) lo = mid; else hi = mid;
    }
    // sequences without blocks share ibase with their successor: the search lands on the last
    // sequence whose ibase <= i, which is the one that owns block i
    BlockDesc d;
    d.soff = (uint64_t)seq_off[lo];
    d.len = (uint32_t)(seq_off[lo + 1] - seq_off[lo]);
    d.j = i - ibase[lo];
    d.nk = ibase[lo + 1] - ibase[lo];
    d.ibase = ibase[lo];
    d.seq = lo;
    d.pad = 0;
    out[i] = d;
}

// ---------------------------------------------------------------------------------------
// Exclusive prefix sum over uint32 (three launches: partial sums, scan of partials, local scan).
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanChunk = kScanThreads * kScanPerThread;

__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t x, uint32_t *total, uint32_t *lds /*>= 4 + 1*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = x;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
    for (int w = 0; w < kScanThreads / 64; w++) {
        uint32_t s = lds[w];
        if (w < wave) wbase += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return wbase + incl - x;
}

__global__ __launch_bounds__(kScanThreads) void scan_partials_kernel(const uint32_t *in, uint64_t n, uint64_t *partial)
{
    __shared__ uint32_t lds[8];
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t s = 0;
    for (int k = 0; k < kScanPerThread; k++)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    wg_exclusive_scan(s, &total, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// single workgroup: partial[] -> exclusive (in place), grand total to *total_out
__global__ __launch_bounds__(kScanThreads) void scan_top_kernel(uint64_t *partial, uint32_t n_partials, uint64_t *total_out)
{
    __shared__ uint64_t carry;
    __shared__ uint64_t wsum[kScanThreads / 64];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = 0; b < n_partials; b += kScanThreads) {
        uint32_t i = b + threadIdx.x;
        uint64_t x = i < n_partials ? partial[i] : 0;
        uint64_t incl = x;
        for (int off = 1; off < 64; off <<= 1) {
            uint64_t y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t wbase = 0, all = 0;
        for (int w = 0; w < kScanThreads / 64; w++) {
            uint64_t s = wsum[w];
            if (w < wave) wbase += s;
            all += s;
        }
        uint64_t c = carry;
        if (i < n_partials) partial[i] = c + wbase + incl - x;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kScanThreads) void scan_final_kernel(const uint32_t *in, uint64_t n, const uint64_t *partial,
                                                                  uint32_t *out)
{
    __shared__ uint32_t lds[8];
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread];
    uint32_t s = 0;
    for (int k = 0; k < kScanPerThread; k++) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t total;
    uint32_t excl = wg_exclusive_scan(s, &total, lds);
    uint32_t run = (uint32_t)partial[blockIdx.x] + excl;
    for (int k = 0; k < kScanPerThread; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------
// Ordered placement: staging (block-granular, arbitrary block order) -> hits[] ordered by
// (container, from0InProt).  One wave per block; rows in the order the scan kernel staged them.
template <bool AA>
__global__ __launch_bounds__(kWave *kWavesPerWG) void place_kernel(const BlockDesc *blocks, uint32_t n_blocks,
                                                                  const uint32_t *counts, const uint32_t *offs,
                                                                  const uint32_t *block_stage_base,
                                                                  const kg_hit *stage, kg_hit *hits)
{
    constexpr int ROWS = AA ? 1 : 6;
    const int lane = threadIdx.x & 63;
    const uint32_t it = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerWG + (threadIdx.x >> 6));
    if (it >= n_blocks) return;
    const BlockDesc bd = blocks[it];
    uint32_t src = block_stage_base[it];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        uint32_t vrow;
        if (AA) vrow = it;
        else if (r < 3) vrow = 6u * bd.ibase + (uint32_t)r * bd.nk + bd.j;
        else vrow = 6u * bd.ibase + (3u + (bd.len - (uint32_t)(r - 3)) % 3u) * bd.nk + (bd.nk - 1u - bd.j);
        uint32_t n = counts[vrow];
        if ((uint32_t)lane < n) hits[(uint64_t)offs[vrow] + lane] = stage[(uint64_t)src + lane];
        src += n;
    }
}

// container_hit_start[c] for every container, plus the end sentinel
template <bool AA>
__global__ void container_starts_kernel(const uint32_t *ibase, uint32_t n_seqs, const uint32_t *offs, uint64_t n_rows,
                                        const uint64_t *total, int64_t *chs)
{
    constexpr uint32_t PER = AA ? 1 : 6;
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t n_cont = (uint64_t)n_seqs * PER;
    if (c > n_cont) return;
    if (c == n_cont) { chs[c] = (int64_t)*total; return; }
    uint32_t k = (uint32_t)(c / PER), cc = (uint32_t)(c % PER);
    uint32_t nk = ibase[k + 1] - ibase[k];
    uint64_t row = (uint64_t)PER * ibase[k] + (uint64_t)cc * nk;
    chs[c] = row < n_rows ? (int64_t)offs[row] : (int64_t)*total;
}

// ---------------------------------------------------------------------------------------
// Aggregation.  gatherHits (KGJ:457-514) + processSetOfHits (KGJ:385-455) as a state machine over
// the container's position-ordered hits.  The reference's "hits" list is always the accepted
// records inside one index range [lo, last] of that array (it is only ever cleared or cut down to
// its last two members), so the list is represented by (lo, last, prev, cnt) plus one "accepted"
// byte per hit (order constraint, KGJ:490-494, and the 39 998 cap, KGJ:496, reject records).
struct AggParams { int32_t min_hits, min_weighted_hits, max_gap, order_constraint; };

struct CallSpan { uint32_t lo, last_hit; };   // global hit indices of the called set's first record and last voter

template <bool EMIT>
struct CallSink {
    kg_call *calls; CallSpan *spans; uint64_t at; uint32_t n;
};

template <bool EMIT>
__device__ void gather_container(const kg_hit *h, int64_t begin, int64_t end, const AggParams p, uint8_t *acc,
                                 uint32_t container, CallSink<EMIT> &sink)
{
    int64_t lo = begin, last = begin, prev = begin;
    int32_t cnt = 0;
    int32_t currentFI = 0;

    auto process = [&]() {
        // KGJ:387-396
        int32_t fICount = 0;
        float weighted = 0.f;
        int64_t lastHit = lo;
        for (int64_t k = lo; k <= last; k++) {
            if (acc[k] && h[k].fI == currentFI) {
                lastHit = k;
                fICount++;
                weighted += h[k].functionWt;      // float32, list order
            }
        }
        if (fICount >= p.min_hits && weighted >= (float)p.min_weighted_hits) {   // KGJ:397
            if (EMIT) {
                kg_call c;
                c.container = container;
                c.start = h[lo].from0InProt;
                c.end = h[lastHit].from0InProt + (KG_K - 1);
                c.count = fICount; c.fI = currentFI; c.weightedHits = weighted;
                sink.calls[sink.at + sink.n] = c;
                CallSpan s; s.lo = (uint32_t)lo; s.last_hit = (uint32_t)lastHit;
                sink.spans[sink.at + sink.n] = s;
            }
            sink.n++;
        }
        // KGJ:441-453: keep the last two records if they start a new function, else clear
        if (h[prev].fI != currentFI && h[prev].fI == h[last].fI) {
            currentFI = h[last].fI;
            lo = prev;
            cnt = 2;
        } else {
            cnt = 0;
        }
    };

    for (int64_t i = begin; i < end; i++) {
        const int32_t ppos = h[i].from0InProt, fI = h[i].fI, avg = h[i].avgOffFromEnd;
        if (cnt > 0 && (int32_t)((uint32_t)h[last].from0InProt + (uint32_t)p.max_gap) < ppos) {   // KGJ:477-484
            if (cnt >= p.min_hits) process(); else cnt = 0;
        }
        if (cnt == 0) currentFI = fI;                                                             // KGJ:486-488
        bool ok = !p.order_constraint || cnt == 0;
        if (!ok) {                                                                                // KGJ:490-494
            int32_t d = (int32_t)((uint32_t)(ppos - h[last].from0InProt) - (uint32_t)(h[last].avgOffFromEnd - avg));
            int32_t ad = d < 0 ? (int32_t)(0u - (uint32_t)d) : d;      // Math.abs(int)
            ok = fI == h[last].fI && ad <= 20;
        }
        bool appended = false;
        if (ok) {
            if (cnt < KG_MAX_HITS_PER_SEQ - 2) {                                                  // KGJ:496-497
                if (cnt == 0) { lo = i; prev = i; } else { prev = last; }
                last = i;
                cnt++;
                appended = true;
            }
        }
        acc[i] = appended ? 1 : 0;
        if (ok && cnt > 1 && currentFI != fI && h[prev].fI == h[last].fI) process();             // KGJ:503-508
    }
    if (cnt >= p.min_hits) process();                                                            // KGJ:511-513
}

// one lane per container
template <bool EMIT>
__global__ void calls_kernel(const kg_hit *hits, const int64_t *chs, uint64_t n_cont, AggParams p, uint8_t *acc,
                             uint32_t *call_cnt, const uint32_t *call_off, kg_call *calls, CallSpan *spans)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cont) return;
    CallSink<EMIT> sink;
    sink.calls = calls; sink.spans = spans; sink.n = 0;
    sink.at = EMIT ? call_off[c] : 0;
    gather_container<EMIT>(hits, chs[c], chs[c + 1], p, acc, (uint32_t)c, sink);
    if (!EMIT) call_cnt[c] = sink.n;
}

// ccs[c] = call_off[c] widened, plus sentinel
__global__ void call_starts_kernel(const uint32_t *call_off, uint64_t n_cont, const uint64_t *total, int64_t *ccs)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cont) return;
    ccs[c] = c == n_cont ? (int64_t)*total : (int64_t)call_off[c];
}

// OTU vote (KGJ:413-439), one lane per sequence: replay the voters of every CALL of the sequence
// in emission order against the 5-entry buffer that persists across the sequence's containers
// (KGJ:528, 540).
__global__ void otu_kernel(const kg_hit *hits, const uint8_t *acc, const kg_call *calls, const CallSpan *spans,
                           const int64_t *ccs, uint32_t n_seqs, uint32_t per, kg_otu *otu)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    int32_t n = 0;
    int32_t cnt[KG_OI_BUFSZ] = {0, 0, 0, 0, 0}, oi[KG_OI_BUFSZ] = {0, 0, 0, 0, 0};
    int64_t c0 = ccs[(uint64_t)s * per], c1 = ccs[(uint64_t)(s + 1) * per];
    for (int64_t c = c0; c < c1; c++) {
        const int32_t fI = calls[c].fI;
        const CallSpan sp = spans[c];
        for (uint32_t k = sp.lo; k <= sp.last_hit; k++) {
            if (!acc[k] || hits[k].fI != fI) continue;
            const int32_t o = hits[k].oI;
            int j = 0;
            while (j < n && oi[j] != o) j++;
            if (j == n) {
                if (n == KG_OI_BUFSZ) j--; else n++;
                oi[j] = o; cnt[j] = 1;
            } else {
                cnt[j]++;
            }
            while (j > 0 && cnt[j - 1] <= cnt[j]) {
                int32_t tc = cnt[j - 1], to = oi[j - 1];
                cnt[j - 1] = cnt[j]; oi[j - 1] = oi[j];
                cnt[j] = tc; oi[j] = to;
                j--;
            }
        }
    }
    kg_otu r;
    r.n = n;
    for (int k = 0; k < KG_OI_BUFSZ; k++) { r.count[k] = k < n ? cnt[k] : 0; r.oI[k] = k < n ? oi[k] : 0; }
    otu[s] = r;
}

// ---------------------------------------------------------------------------------------
// tag array from the 24-byte records (one pass over the table at load time)
__global__ void build_tags_kernel(const uint8_t *entries, uint64_t limit, uint64_t n_tags, uint8_t *tags,
                                  unsigned long long *occupied)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long occ = 0;
    for (; i < n_tags; i += stride) {
        uint32_t t = kTagEmpty;
        if (i < limit) {
            const uint2 *p = reinterpret_cast<const uint2 *>(entries + i * 24);
            uint2 a = p[0];
            int64_t key = (int64_t)(((uint64_t)a.y << 32) | a.x);
            if (key <= KG_MAX_ENCODED) {        // occupied (KGJ:1000); negative keys are occupied and never match
                t = tag_of((uint64_t)key);
                occ++;
            }
        }
        tags[i] = (uint8_t)t;
    }
    for (int off = 32; off > 0; off >>= 1) occ += __shfl_down(occ, off);
    if ((threadIdx.x & 63) == 0 && occ) atomicAdd(occupied, occ);
}

}  // namespace kg
