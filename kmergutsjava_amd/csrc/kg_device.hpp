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
    uint32_t m35;        // floor(2^35 / num_sigs) when 64 <= num_sigs < 2^31 (split_fast applies), else 0
};

// The wave's issue priority among the waves of its SIMD (0..3; the instruction takes an immediate).  The kernels that share the CUs
// in the partitioned pipeline set it per phase: a wave that issues a few instructions between long waits (LDS round trips of the
// scatter pass's insert phase, L2 round trips of the index pass) should not queue behind other waves' VALU streams.
__device__ __forceinline__ void set_wave_prio(uint32_t p)
{
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
    // LDS traffic of one wave is executed in issue order; this only stops the compiler from
    // moving LDS reads of other lanes' data above the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A query k-mer is handled as (q, home slot) with value = q * num_sigs + slot (KGJ:969: slot = value % numSigs).
//
// split_value: exact quotient / remainder of any v < 2^63 (one correction step suffices: q_est in {q-1, q}).
__device__ __forceinline__ uint64_t split_value(uint64_t v, const TableView &t, uint64_t *q_out)
{
    uint64_t q = __umul64hi(v, t.magic);
    uint64_t r = v - q * t.num_sigs;
    if (r >= t.num_sigs) { r -= t.num_sigs; q += 1; }
    *q_out = q;
    return r;
}

// split_fast: the same for value = hi * 160000 + lo (hi, lo < 160000: the two 4-residue half codes) when
// 64 <= n < 2^31, with 24-bit multiplies (full rate) and two 32-bit ones instead of 64-bit arithmetic.
//   vh = value >> 3 exactly (hi * 160000 is a multiple of 8), < 2^32;  m35 = floor(2^35 / n)
//   q_est = floor(vh * m35 / 2^32) <= value / n, and value / n - vh * m35 / 2^32 < value / 2^35 + 8 / n < 0.75 + 0.125,
//   so q_est is floor(value / n) or one less, the remainder estimate is < 2n < 2^32 and 32-bit arithmetic is exact.
__device__ __forceinline__ uint32_t split_fast(uint32_t hi, uint32_t lo, uint32_t n, uint32_t m35, uint32_t *q_out)
{
    const uint32_t vh = __umul24(hi, 20000u) + (lo >> 3);
    const uint32_t v32 = __umul24(hi, 160000u) + lo;        // value mod 2^32
    uint32_t q = __umulhi(vh, m35);
    uint32_t r = v32 - q * n;
    if (r >= n) { r -= n; q += 1; }
    *q_out = q;
    return r;
}

// the two half codes of a k-mer -> (q, slot); uniform choice of the arithmetic
__device__ __forceinline__ uint64_t split_halves(uint32_t hi, uint32_t lo, const TableView &t, uint64_t *q_out)
{
    if (t.m35) {
        uint32_t q;
        const uint32_t r = split_fast(hi, lo, (uint32_t)t.num_sigs, t.m35, &q);
        *q_out = q;
        return r;
    }
    return split_value((uint64_t)hi * 160000ull + lo, t, q_out);
}

// 8-bit fingerprint of a k-mer, from its (q, slot) form: 24-bit multiplies only; the byte is taken from the
// middle of the products, where every low input bit has spread.  Keys that share a home slot differ in q and
// never collide; neighbours collide at the ideal 1/255 (tools/notes in profiles/r01_partition_path.md).
// (as the instruction: in the probe loops the compiler turns __umul24 by a constant into v_mul_lo_u32, quarter rate)
__device__ __forceinline__ uint32_t mul24(uint32_t x, uint32_t c)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(c), "v"(x));
    return r;
}

__device__ __forceinline__ uint32_t tag_qs(uint64_t q, uint64_t slot)
{
    const uint32_t q32 = (uint32_t)q ^ (uint32_t)(q >> 32);
    const uint32_t h = mul24((uint32_t)slot, 0x9E3779u) ^ mul24((uint32_t)(slot >> 24), 0x85EBCBu) ^
                       mul24(q32 ^ (q32 >> 19), 0xC2B2AFu);
    const uint32_t t = (h >> 16) & 0xFFu;
    return t == kTagEmpty ? 0xFEu : t;
}

// home slot only
__device__ __forceinline__ uint64_t home_slot(uint64_t v, const TableView &t)
{
    uint64_t q;
    return split_value(v, t, &q);
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

// First slot i in skip..15 whose tag is EMPTY or == fp (bytes below `skip` are ignored).  Returns 16 if
// none.  *is_empty tells which.
__device__ __forceinline__ int first_stop(const Tags16 &x, uint32_t fp, bool *is_empty, uint32_t skip = 0)
{
    const uint32_t fpw = __builtin_amdgcn_perm(0u, fp, 0u);     // fp (one byte) in all four bytes
    uint32_t e0 = zero_bytes(~x.w[0]), e1 = zero_bytes(~x.w[1]), e2 = zero_bytes(~x.w[2]), e3 = zero_bytes(~x.w[3]);
    uint32_t f0 = zero_bytes(x.w[0] ^ fpw), f1 = zero_bytes(x.w[1] ^ fpw), f2 = zero_bytes(x.w[2] ^ fpw), f3 = zero_bytes(x.w[3] ^ fpw);
    uint64_t elo = ((uint64_t)e1 << 32) | e0, ehi = ((uint64_t)e3 << 32) | e2;
    uint64_t slo = elo | (((uint64_t)f1 << 32) | f0), shi = ehi | (((uint64_t)f3 << 32) | f2);
    if (skip) {
        // keep only the stop bits of bytes >= skip
        const uint64_t keep_lo = skip >= 8 ? 0ull : ~0ull << (8 * skip);
        const uint64_t keep_hi = skip <= 8 ? ~0ull : ~0ull << (8 * (skip - 8));
        slo &= keep_lo;
        shi &= keep_hi;
    }
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

// A 16-byte window starting at `slot` straddles a 128-byte line (= one more L2-miss request, the unit the
// memory side fetches) when slot % 128 > 112.  Then read the aligned 16-byte chunk that holds `slot` instead
// and ignore its first slot % 16 bytes: fewer slots of look-ahead in 12 % of the probes, 10 % fewer requests.
__device__ __forceinline__ uint64_t probe_window(uint64_t slot, uint32_t *skip)
{
    const bool straddles = ((uint32_t)slot & 127u) > 112u;
    *skip = straddles ? ((uint32_t)slot & 15u) : 0u;
    return straddles ? (slot & ~15ull) : slot;
}

struct Entry { int64_t key; int32_t oI, avg, fI; float wt; };
struct Payload { int32_t oI, avg, fI; float wt; };

__device__ __forceinline__ Entry load_entry(const TableView &t, uint64_t slot)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(t.entries + slot * 24);   // 8-byte aligned
    uint2 a = p[0], b = p[1], c = p[2];
    Entry e;
    e.key = (int64_t)(((uint64_t)a.y << 32) | a.x);
    e.oI = (int32_t)b.x; e.avg = (int32_t)b.y; e.fI = (int32_t)c.x; e.wt = __uint_as_float(c.y);
    return e;
}

// Streaming accesses: lists that are written once and read once, a pass later, from HBM carry the non-temporal hint so
// that they do not push the tag pass's L2-resident tags (and each other) out of the 4 MiB L2 of their XCD.
typedef uint32_t kg_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t kg_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store16(void *dst, const void *src16)
{
    kg_u32x4 v;
    __builtin_memcpy(&v, src16, 16);
    __builtin_nontemporal_store(v, reinterpret_cast<kg_u32x4 *>(dst));
}
__device__ __forceinline__ void stream_store8(void *dst, uint64_t w)
{
    kg_u32x2 v; v.x = (uint32_t)w; v.y = (uint32_t)(w >> 32);
    __builtin_nontemporal_store(v, reinterpret_cast<kg_u32x2 *>(dst));
}
__device__ __forceinline__ void stream_store_hit(kg_hit *dst, const kg_hit &h)      // 24 bytes, 8-byte aligned
{
    kg_u32x4 a; a.x = h.container; a.y = (uint32_t)h.from0InProt; a.z = (uint32_t)h.oI; a.w = (uint32_t)h.avgOffFromEnd;
    kg_u32x2 b; b.x = (uint32_t)h.fI; b.y = __float_as_uint(h.functionWt);
    __builtin_nontemporal_store(a, reinterpret_cast<kg_u32x4 *>(dst));
    __builtin_nontemporal_store(b, reinterpret_cast<kg_u32x2 *>(reinterpret_cast<unsigned char *>(dst) + 16));
}
__device__ __forceinline__ kg_hit stream_load_hit(const kg_hit *src)
{
    const kg_u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const kg_u32x4 *>(src));
    const kg_u32x2 b = __builtin_nontemporal_load(reinterpret_cast<const kg_u32x2 *>(reinterpret_cast<const unsigned char *>(src) + 16));
    kg_hit h;
    h.container = a.x; h.from0InProt = (int32_t)a.y; h.oI = (int32_t)a.z; h.avgOffFromEnd = (int32_t)a.w;
    h.fI = (int32_t)b.x; h.functionWt = __uint_as_float(b.y);
    return h;
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
__device__ __forceinline__ uint32_t aa_code_of_rt(char c)
{
    uint32_t r = 20;
#pragma unroll
    for (uint32_t i = 0; i < 20; i++)
        if ("ACDEFGHIKLMNPQRSTVWY"[i] == c) r = i;
    return r;
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

// The encode stage is table driven and branch free, and works on BYTES: a residue / codon code is one byte, 0..19 for
// an amino acid (toAminoAcidOff, KGJ:111-175) and kBadByte for everything else (code 20: stop codons, codons with a
// non-ACGTU base, non-residue characters).  The eight codes of a window are eight consecutive bytes of LDS, fetched
// with ONE (unaligned) 8-byte read; v_dot4_u32_u8 folds two of them at a time:
//     half code = c0 * 8000 + c1 * 400 + c2 * 20 + c3 = (c0 * 20 + c1) * 400 + (c2 * 20 + c3)      (< 160000)
// and a window is valid iff none of its bytes has the kBadByte bit (encodedKmer's early return, KGJ:283-285).
// Per block and lane this is ~27 LDS instructions (the dword-per-code version it replaces needed ~88, and 3.7 KB of
// LDS per wave instead of 0.7: the scatter pass is LDS-bound, profiles/r02_pipeline.md section 3).
constexpr uint32_t kBadByte = 0x80u;
__device__ __forceinline__ uint32_t dot4(uint32_t bytes, uint32_t weights, uint32_t acc = 0u)
{
    return __builtin_amdgcn_udot4(bytes, weights, acc, false);
}
// Eight (or four) consecutive bytes of LDS at an arbitrary byte address (base 4-byte aligned).
// KG_ENC_UNALIGNED: one unaligned ds_read_b64 / b32 (gfx950 serves them, tools/lds_unaligned.hip); otherwise the
// aligned dwords around the address and v_alignbyte.
__device__ __forceinline__ uint2 lds_bytes8(const uint8_t *base, uint32_t at)
{
    uint2 v;
#ifdef KG_ENC_UNALIGNED
    __builtin_memcpy(&v, base + at, 8);
#else
    const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (at & ~3u));
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], sh = at & 3u;
    v.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
    v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
#endif
    return v;
}
__device__ __forceinline__ uint32_t lds_bytes4(const uint8_t *base, uint32_t at)
{
#ifdef KG_ENC_UNALIGNED
    uint32_t x;
    __builtin_memcpy(&x, base + at, 4);
    return x;
#else
    const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (at & ~3u));
    return __builtin_amdgcn_alignbyte(p[1], p[0], at & 3u);
#endif
}

// the 48-bit product of two 24-bit numbers, both halves at full rate.  (Inline asm: hipcc's hazard recognizer does not see
// what an asm statement reads, so the operands must not come straight out of a v_dot4 / MFMA / transcendental -- those need
// wait states before a dependent VALU; round 4 learnt it from an asm'd v_mad_u32_u24 behind two v_dot4: wrong k-mer codes.)
__device__ __forceinline__ uint64_t mul24_wide(uint32_t a, uint32_t b)
{
    uint32_t lo, hi;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    return ((uint64_t)hi << 32) | lo;
}
// bytes (c0, c1, c2, c3) of w, c0 lowest -> c0 * 8000 + c1 * 400 + c2 * 20 + c3
__device__ __forceinline__ uint32_t half_up(uint32_t w)
{
    // (the second dot accumulates onto the first one's product: written as product + dot, hipcc emits a quarter-rate
    //  v_mad_u64_u32 for the half that feeds split_fast's 24-bit multiplies -- one per window row in the scatter pass)
    return dot4(w, 0x01140000u, __umul24(dot4(w, 0x00000114u), 400u));
}
// the same with c3 the most significant: c3 * 8000 + c2 * 400 + c1 * 20 + c0
__device__ __forceinline__ uint32_t half_down(uint32_t w)
{
    return dot4(w, 0x00001401u, __umul24(dot4(w, 0x14010000u), 400u));
}

// Lookup tables shared by the waves of a workgroup (built once per workgroup by encode_init).
struct EncTablesDna {
    uint16_t codon[128];  // [b0*25 + b1*5 + b2] (base codes 0..4): low byte = code of codon b0 b1 b2 (translate, KGJ:320-343),
                          // high byte = code of the reverse-complement codon compl(b2) compl(b1) compl(b0) (KGJ:263-272)
    uint8_t base[256];    // dnaChar
};
struct EncTablesAa {
    uint8_t code[256];    // toAminoAcidOff (KGJ:111-175): 0..19, else kBadByte
};

struct __attribute__((aligned(16))) WaveLdsDna {
    uint8_t bc[240];      // base codes of the block's (up to) 216 bases
    uint8_t cf[3][80];    // cf[f][t]: code of the forward codon starting at base 3t + f            (t <= 70)
    uint8_t cr[3][80];    // cr[f][t]: code of the reverse-complement codon over bases 3t + f .. 3t + f + 2
};

struct __attribute__((aligned(16))) WaveLdsAa {
    uint8_t code[80];     // residue codes of the block's (up to) 71 characters
};

template <bool AA> struct WaveLds;
template <> struct WaveLds<false> { typedef WaveLdsDna type; typedef EncTablesDna tables; };
template <> struct WaveLds<true> { typedef WaveLdsAa type; typedef EncTablesAa tables; };

// once per workgroup (all threads call; the caller synchronises the workgroup afterwards)
template <bool AA>
__device__ __forceinline__ void encode_init(typename WaveLds<AA>::tables &t, uint32_t tid, uint32_t n_threads)
{
    if constexpr (AA) {
        for (uint32_t b = tid; b < 256; b += n_threads) {
            const uint32_t c = aa_code_of_rt((char)b);
            t.code[b] = (uint8_t)(c < 20 ? c : kBadByte);
        }
    } else {
        for (uint32_t b = tid; b < 256; b += n_threads) t.base[b] = (uint8_t)dna_code(b);
        for (uint32_t i = tid; i < 128; i += n_threads) {
            const uint32_t b0 = i / 25u, b1 = (i / 5u) % 5u, b2 = i % 5u;
            uint32_t f = kBadByte, r = kBadByte;
            if (i < 125 && b0 < 4 && b1 < 4 && b2 < 4) {
                // codon index c1*16+c2*4+c3 (KGJ:331-337); kCodon16[c1*4+c2] packs the four c3 codes
                f = (kCodon16[b0 * 4 + b1] >> (b2 * 5)) & 31u;
                r = (kCodon16[(3 - b2) * 4 + (3 - b1)] >> ((3 - b0) * 5)) & 31u;   // compl code = 3 - code
                if (f >= 20) f = kBadByte;                                         // stop codon '*' -> code 20
                if (r >= 20) r = kBadByte;
            }
            t.codon[i] = (uint16_t)(f | (r << 8));
        }
    }
}

// The raw characters of one block.  DNA: raw[0] = the four characters 4 * lane .. 4 * lane + 3 of the block's
// (up to) 216-character window, one (unaligned) dword load per lane, 'N' behind the end of the sequence (never read
// from memory: the next sequence, or the end of the caller's buffer, lies there); AA: raw[0], raw[1] = characters
// lane and lane + 64.  Separate from the encode step so that a caller can fetch the next block's characters while it
// works on the current one.
template <bool AA>
__device__ __forceinline__ void load_block_chars(const uint8_t *__restrict__ seq, const BlockDesc &bd, int lane, uint32_t (&raw)[4])
{
    const uint64_t soff = bd.soff;
    if constexpr (AA) {
        const uint32_t w0 = bd.j * kAaWinPerBlock;
        const uint32_t nload = min(71u, bd.len - w0);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t q = (uint32_t)lane + 64u * k;
            raw[k] = q < nload ? seq[soff + w0 + q] : 0u;
        }
        raw[2] = raw[3] = 0;
    } else {
        const uint32_t ts = bd.j * kDnaPosPerBlock;
        const uint32_t nload = min(216u, bd.len - ts);      // 215 are needed (192 + 23); the 216th completes the last dword
        const uint32_t at = 4u * (uint32_t)lane;
        uint32_t w = 0x4E4E4E4Eu;                           // "NNNN"
        const uint8_t *p = seq + soff + ts + at;
        if (at + 4u <= nload) {
            __builtin_memcpy(&w, p, 4);                     // one global_load_dword at byte alignment
        } else if (at < nload) {                            // the last characters of the sequence (one lane per sequence)
            for (uint32_t k = 0; at + k < nload; k++) w = (w & ~(0xFFu << (8u * k))) | ((uint32_t)p[k] << (8u * k));
        }
        raw[0] = w;
        raw[1] = raw[2] = raw[3] = 0;
    }
}

// Leave the residue / codon codes of one block in LDS (all lanes of the wave), from its raw characters.
template <bool AA>
__device__ __forceinline__ void encode_chars(typename WaveLds<AA>::type &l, const typename WaveLds<AA>::tables &t,
                                             const uint32_t (&raw)[4], int lane)
{
    if constexpr (AA) {
        // ---- protein: windows i = 64j + lane
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t q = (uint32_t)lane + 64u * k;
            const uint8_t c = t.code[raw[k] & 255u];
            if (k < 1 || q < 72) l.code[q] = c;
        }
        wave_sync();
    } else {
        // ---- DNA: 216 bases -> base codes (four per lane, one dword store), then the codon codes of both strands
        {
            const uint32_t w = raw[0];
            const uint32_t c = (uint32_t)t.base[w & 255u] | ((uint32_t)t.base[(w >> 8) & 255u] << 8) |
                               ((uint32_t)t.base[(w >> 16) & 255u] << 16) | ((uint32_t)t.base[w >> 24] << 24);
            if (lane < 60) *reinterpret_cast<uint32_t *>(&l.bc[4 * lane]) = c;
        }
        wave_sync();
        // codon q = lane + 64k (q <= 212): its three base codes are three bytes of one (unaligned) dword read, the table
        // index b0*25 + b1*5 + b2 is one v_dot4; all four reads are in flight together (clamped reads, predicated writes)
        {
            uint32_t idx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q = min((uint32_t)lane + 64u * k, 212u);
                idx[k] = dot4(lds_bytes4(l.bc, q), 0x00010519u);
            }
            uint32_t cw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cw[k] = t.codon[idx[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q = (uint32_t)lane + 64u * k;
                const uint32_t tt = (q * 171u) >> 9;         // q / 3 for q < 256
                const uint32_t f = q - 3u * tt;
                if (k < 3 || q < 213) {
                    l.cf[0][f * 80u + tt] = (uint8_t)cw[k];
                    l.cr[0][f * 80u + tt] = (uint8_t)(cw[k] >> 8);
                }
            }
        }
        wave_sync();
    }
}

template <bool AA>
__device__ __forceinline__ void encode_block(typename WaveLds<AA>::type &l, const typename WaveLds<AA>::tables &t,
                                             const uint8_t *__restrict__ seq, const BlockDesc &bd, int lane)
{
    uint32_t raw[4];
    load_block_chars<AA>(seq, bd, lane, raw);
    encode_chars<AA>(l, t, raw, lane);
}

// encodedKmer (KGJ:274-292) of the lane's window in row r (wave-uniform; DNA: strand r/3, phase r%3), as its
// two half codes: value = hi * 160000 + lo.  Returns whether the window is a query k-mer.
template <bool AA>
__device__ __forceinline__ bool row_halves(const typename WaveLds<AA>::type &l, int r, int lane, const BlockDesc &bd,
                                           uint32_t *hi_out, uint32_t *lo_out)
{
    if constexpr (AA) {
        const uint2 v = lds_bytes8(l.code, (uint32_t)lane);   // the window's eight residue codes
        const uint32_t i = bd.j * kAaWinPerBlock + lane;
        *hi_out = half_up(v.x); *lo_out = half_up(v.y);
        // queried iff i < len - 8 (KGJ:912: i < pIseq.length - K -- the last window is never queried)
        return (((v.x | v.y) & 0x80808080u) == 0u) & ((uint64_t)i + 8 < (uint64_t)bd.len);
    } else {
        const bool minus = r >= 3;
        const uint32_t f = (uint32_t)(minus ? r - 3 : r);
        // window over forward bases p .. p+23, p = 3 * lane + f: its eight codons are the bytes lane .. lane+7 of phase f
        const uint2 v = lds_bytes8(minus ? &l.cr[0][0] : &l.cf[0][0], f * 80u + (uint32_t)lane);
        if (minus) {
            // on the '-' strand the codon over the highest bases comes first (KGJ:263-272, 1068-1072)
            *hi_out = half_down(v.y); *lo_out = half_down(v.x);
        } else {
            *hi_out = half_up(v.x); *lo_out = half_up(v.y);
        }
        return ((v.x | v.y) & 0x80808080u) == 0u;
    }
}

// Index of row r of block bd in the container-major row order (rows of one container are contiguous), the
// container it belongs to and the residue index of lane 0's window / its direction.
template <bool AA>
__device__ __forceinline__ uint32_t row_index(const BlockDesc &bd, uint32_t it, int r)
{
    if (AA) return it;
    const uint32_t vbase = 6u * bd.ibase;
    if (r < 3) return vbase + (uint32_t)r * bd.nk + bd.j;
    return vbase + (3u + (bd.len - (uint32_t)(r - 3)) % 3u) * bd.nk + (bd.nk - 1u - bd.j);
}

template <bool AA>
__device__ __forceinline__ void row_record_key(const BlockDesc &bd, int r, int lane, uint32_t *container, int32_t *pos)
{
    if (AA) {
        *container = bd.seq;
        *pos = (int32_t)(bd.j * kAaWinPerBlock + lane);
    } else if (r < 3) {
        // '+' strand: frame r (block start is a multiple of 3), residue index 64j + lane
        *container = bd.seq * 6u + (uint32_t)r;
        *pos = (int32_t)(bd.j * 64u + lane);
    } else {
        // '-' strand: the window over forward bases p..p+23 starts at reverse-complement base b' = L-24-p,
        // i.e. frame b'%3, residue b'/3 (KGJ:1068-1072).  24 % 3 == 0 and the block start is a multiple of 3,
        // so the frame depends on the row only.
        const uint32_t f = (uint32_t)(r - 3);
        *container = bd.seq * 6u + 3u + (bd.len - f) % 3u;
        *pos = (int32_t)((bd.len - 24u - f - bd.j * kDnaPosPerBlock) / 3u) - (int32_t)lane;
    }
}

// ---------------------------------------------------------------------------------------
// KG_F_PROGRESS: what the reference's table stream reports while its merge-join runs (KGJ:1016-1025: a "Processed: NN%" line
// whenever the tenth of the table changes -- at slots the join VISITS, i.e. slots some query's walk reads) and where it fails on
// a table file shorter than numSigs records (KGJ:985-988, 1036-1049).  The walking kernels note every query's walk
// [home slot, last slot read] here; the host turns it into the lines (kmer_guts_java.py, kmer_guts_cli.cpp).
struct Progress {
    unsigned long long first[11];     // first[d]: smallest slot visited in tenth d (~0: none); tenth of slot s = bounds below
    unsigned long long last_plus1;    // 1 + the largest slot visited (0: none)
    unsigned long long first_beyond;  // smallest home slot >= limit among the query k-mers (~0: none): the stream ends before it
    unsigned long long walk_ran_off;  // 1: some walk reached the end of the record stream undecided (EOFException, KGJ:1097-1126)
    unsigned long long lo[11];        // lo[d]: smallest slot whose tenth is >= d (host-computed with the reference's double
                                      // arithmetic, KGJ:1018); lo[0] = 0, ~0 when no slot reaches d
    unsigned long long found_upto[11];// distinct k-mers found at slots <= first[d] (count_found_kernel): "found-so-far"
    unsigned long long kmers_found;   // distinct k-mers found = distinct slots among the hit records (KGJ:1004-1006)
};

// kmersFound: a k-mer counts once however many query positions carry it (KGJ:1004-1015), and a k-mer is found at one slot:
// the distinct slots of the hit records, marked in a bitmap over the table's slots and counted up to each tenth's first
// visited slot.
__global__ void mark_found_kernel(const uint32_t *__restrict__ slots, uint64_t n, uint32_t *bitmap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t s = slots[i];
        atomicOr(&bitmap[s >> 5], 1u << (s & 31u));
    }
}
__global__ void count_found_kernel(const uint32_t *__restrict__ bitmap, uint64_t n_words, Progress *p)
{
    unsigned long long upto[11], all = 0;
    for (int f = 0; f <= 10; f++) upto[f] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint32_t bits = bitmap[w];
        if (!bits) continue;
        all += (unsigned long long)__popc(bits);
        const uint64_t b = w << 5;
        for (int f = 0; f <= 10; f++) {
            const unsigned long long t = p->first[f];                 // ~0: tenth not visited (its count is not used)
            if (t == ~0ull || t < b) continue;
            const uint64_t k = t - b;                                  // bits 0..k count
            upto[f] += (unsigned long long)__popc(k >= 31 ? bits : bits & ((2u << k) - 1u));
        }
    }
    for (int f = 0; f <= 10; f++) {
        for (int off = 32; off > 0; off >>= 1) upto[f] += __shfl_down(upto[f], off);
        if ((threadIdx.x & 63) == 0 && upto[f]) atomicAdd(&p->found_upto[f], upto[f]);
    }
    for (int off = 32; off > 0; off >>= 1) all += __shfl_down(all, off);
    if ((threadIdx.x & 63) == 0 && all) atomicAdd(&p->kmers_found, all);
}

// One query's walk started at its home slot and stopped at `stop`: the slot of its k-mer or of the first empty record, or
// >= limit when it reached the end of the stream undecided; it read the slots home .. min(stop, limit - 1).  Reads first,
// atomics only for a new minimum / maximum: after the first few walks of a tenth nearly every walk leaves the words alone.
__device__ __forceinline__ void progress_note_walk(Progress *p, uint64_t home, uint64_t stop, uint64_t limit)
{
    const uint64_t last = stop < limit ? stop : limit - 1;
    if (stop >= limit) p->walk_ran_off = 1ull;
    int d = 0;
#pragma unroll
    for (int k = 1; k <= 10; k++) d += p->lo[k] <= home ? 1 : 0;
    if (home < *const_cast<volatile unsigned long long *>(&p->first[d])) atomicMin(&p->first[d], (unsigned long long)home);
    for (int k = d + 1; k <= 10 && p->lo[k] <= last; k++)                       // the walk crosses into tenth k at lo[k]
        if (p->lo[k] < *const_cast<volatile unsigned long long *>(&p->first[k])) atomicMin(&p->first[k], p->lo[k]);
    if (last + 1 > *const_cast<volatile unsigned long long *>(&p->last_plus1)) atomicMax(&p->last_plus1, (unsigned long long)(last + 1));
}
__device__ __forceinline__ void progress_note_beyond(Progress *p, uint64_t home)
{
    if (home < *const_cast<volatile unsigned long long *>(&p->first_beyond)) atomicMin(&p->first_beyond, (unsigned long long)home);
}

// ---------------------------------------------------------------------------------------
// Probe N independent query k-mers per lane (KGJ:944-1034 semantics: from the home slot forward until the
// k-mer, an empty slot or the end of the stream; never wrap).  16 tags per load, records touched only on a
// fingerprint match; the rare longer walks share one copy of the generic walk, rows picked by register muxes.
// On return bit q of the result is set iff query q was found, with ent[q] = the record's payload.
// ran_off is set when a walk reaches the end of the record stream undecided: the point at which the reference's
// table stream throws EOFException and its lookup ends with "Error: null" instead of "Kmers found" (KGJ:799-802,
// 1097-1126); the records are the same either way (EOF == not found).
// val[q] = the k-mer value (compared with the record keys), home_in[q] / fp[q] = its home slot and fingerprint.
template <int N, bool COUNTERS>
__device__ __forceinline__ uint32_t probe_n(const TableView &tab, const uint64_t (&val)[N], const uint64_t (&home_in)[N],
                                            const uint32_t (&fp)[N], bool (&valid)[N], Payload (&ent)[N],
                                            unsigned long long &ctr_valid, unsigned long long &ctr_slots, bool &ran_off,
                                            Progress *prog = nullptr /* COUNTERS only: note the walks (KG_F_PROGRESS) */,
                                            uint32_t *found_slot = nullptr /* COUNTERS only: [N], the slot a found query was found at */)
{
    uint64_t cand[N];     // slot under examination
    uint32_t skip[N];
    Tags16 tg[N];
    uint64_t home[COUNTERS ? N : 1];
#pragma unroll
    for (int q = 0; q < N; q++) {
        cand[q] = home_in[q];
        if (COUNTERS) { home[q] = cand[q]; if (valid[q]) ctr_valid++; }   // query k-mers (KGJ:913-920)
        if (valid[q] && cand[q] >= tab.limit) { ran_off = true; if (COUNTERS && prog) progress_note_beyond(prog, cand[q]); }
        valid[q] = valid[q] && cand[q] < tab.limit;     // beyond the stream: EOF, not found, nothing inspected
        cand[q] = probe_window(cand[q], &skip[q]);
        if (valid[q]) tg[q] = load_tags(tab.tags + cand[q]);
    }
    // state per row: resolved, candidate at cand[q] (bit in st1), or keep walking from cand[q] (bit in pend)
    uint32_t st1 = 0, pend = 0;
    uint64_t stop[COUNTERS ? N : 1];
#pragma unroll
    for (int q = 0; q < N; q++) {
        if (COUNTERS) stop[q] = cand[q];
        if (valid[q]) {
            bool emp;
            int i = first_stop(tg[q], fp[q], &emp, skip[q]);
            if (i == 16) { pend |= 1u << q; cand[q] += 16; }
            else {
                cand[q] += (uint64_t)i;
                if (COUNTERS) stop[q] = cand[q];
                if (!emp) st1 |= 1u << q;
                else if (cand[q] >= tab.limit) ran_off = true;      // the "empty slot" is the padding behind the last record
            }
        }
    }
    uint32_t foundm = 0;
    {
        Entry full[N];
#pragma unroll
        for (int q = 0; q < N; q++)
            if (st1 & (1u << q)) full[q] = load_entry(tab, cand[q]);
#pragma unroll
        for (int q = 0; q < N; q++) {
            if (st1 & (1u << q)) {
                if (full[q].key == (int64_t)val[q]) foundm |= 1u << q;
                else { pend |= 1u << q; cand[q] += 1; }       // fingerprint collision: keep walking
            }
            ent[q].oI = full[q].oI; ent[q].avg = full[q].avg; ent[q].fI = full[q].fI; ent[q].wt = full[q].wt;
        }
    }
    while (__ballot(pend != 0)) {
        if (pend) {
            const int r = __builtin_ctz(pend);
            uint64_t v = val[0], s = cand[0];
            uint32_t f = fp[0];
#pragma unroll
            for (int q = 1; q < N; q++)
                if (r == q) { v = val[q]; s = cand[q]; f = fp[q]; }
            bool done = false, hit = false;
            Entry e;
            e.key = 0; e.oI = e.avg = e.fI = 0; e.wt = 0.f;
            if (s >= tab.limit) { done = true; s = tab.limit; ran_off = true; }
            else {
                Tags16 x = load_tags(tab.tags + s);
                bool emp;
                int i = first_stop(x, f, &emp);
                if (i == 16) s += 16;
                else if (emp) { done = true; s += (uint64_t)i; if (s >= tab.limit) ran_off = true; }
                else {
                    s += (uint64_t)i;
                    e = load_entry(tab, s);
                    if (e.key == (int64_t)v) done = hit = true;
                    else s += 1;
                }
            }
#pragma unroll
            for (int q = 0; q < N; q++) {
                if (r == q) {
                    cand[q] = s;
                    if (done) {
                        if (COUNTERS) stop[q] = s;
                        if (hit) { ent[q].oI = e.oI; ent[q].avg = e.avg; ent[q].fI = e.fI; ent[q].wt = e.wt; foundm |= 1u << q; }
                    }
                }
            }
            if (done) pend &= pend - 1;
        }
    }
    if (COUNTERS) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            if (valid[q]) {
                uint64_t last = stop[q] < tab.limit ? stop[q] + 1 : tab.limit;
                ctr_slots += last - home[q];
                if (prog) progress_note_walk(prog, home[q], stop[q], tab.limit);
                if (found_slot) found_slot[q] = (uint32_t)stop[q];
            }
        }
    }
    return foundm;
}

// ctr[3]: sticky "a probe walked off the end of the record stream" (kg_stats.lookup_ran_off)
__device__ __forceinline__ void flush_ran_off(bool ran_off, unsigned long long *ctr, int lane)
{
    if (__ballot(ran_off) && lane == 0) atomicOr(&ctr[3], 1ull);
}

// wave reduction of the two counters, one atomic pair per wave
__device__ __forceinline__ void flush_counters(unsigned long long ctr_valid, unsigned long long ctr_slots,
                                               unsigned long long *ctr, int lane)
{
    for (int off = 32; off > 0; off >>= 1) {
        ctr_valid += __shfl_down(ctr_valid, off);
        ctr_slots += __shfl_down(ctr_slots, off);
    }
    if (lane == 0) {
        atomicAdd(&ctr[0], ctr_valid);
        atomicAdd(&ctr[1], ctr_slots);
    }
}

// ---------------------------------------------------------------------------------------
// The direct scan kernel.  ROWS = 6 (DNA) or 1 (AA).  Waves are persistent and stride over the blocks.
//
// All pointers are direct kernel arguments (global address space: global_load, counted vmcnt);
// pointers inside a by-value struct would be generic and compile to flat_load + vmcnt(0).
//
// Hit staging: a wave reserves staging records in chunks of `stage_chunk` from one global cursor
// and hands them out to its successive blocks (one returning atomic per ~chunk/hits-per-block
// blocks instead of one per block; a single word sustains only ~90 returning atomics per us).
// The unused tail of a chunk is a hole; block_stage_base[] says where each block's records are.
//
// RPG = rows probed together by one lane (memory-level parallelism per lane vs registers per wave);
// a block's ROWS/RPG row groups are staged independently (block_stage_base has one entry per group).
template <bool AA, bool COUNTERS, int RPG>
__global__ __launch_bounds__(kWave *kWavesPerWG) void scan_kernel(
    const uint8_t *__restrict__ entries, const uint8_t *__restrict__ tags, uint64_t limit, uint64_t num_sigs, uint64_t magic,
    uint32_t m35, const uint8_t *__restrict__ seq, const BlockDesc *__restrict__ blocks, uint32_t n_blocks,
    uint32_t *__restrict__ counts,
    uint32_t *__restrict__ block_stage_base, kg_hit *__restrict__ stage, unsigned long long *cursor, uint64_t stage_cap,
    uint32_t stage_chunk, unsigned long long *ctr,
    Progress *prog /* KG_F_PROGRESS (COUNTERS kernels), else null */, uint32_t *__restrict__ stage_slot /* likewise: parallel to stage[] */,
    const uint32_t *__restrict__ hbits /* one bit per slot (build_hbits_kernel) or null; not read by COUNTERS kernels */, uint64_t tail_start)
{
    constexpr int ROWS = AA ? 1 : 6;
    constexpr int NG = ROWS / RPG;
    static_assert(ROWS % RPG == 0, "RPG must divide the row count");
    __shared__ typename WaveLds<AA>::type lds[kWavesPerWG];
    __shared__ typename WaveLds<AA>::tables enc_tables;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerWG + wave);
    const uint32_t n_waves = gridDim.x * kWavesPerWG;
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic; tab.m35 = m35;

    typename WaveLds<AA>::type &l = lds[wave];
    encode_init<AA>(enc_tables, threadIdx.x, blockDim.x);
    __syncthreads();

    unsigned long long ctr_valid = 0, ctr_slots = 0;
    bool ran_off = false;
    unsigned long long res_at = 0, res_end = 0;          // this wave's staging reservation (uniform)

    for (uint32_t it_v = wave_global; it_v < n_blocks; it_v += n_waves) {
        const uint32_t it = __builtin_amdgcn_readfirstlane(it_v);
        const BlockDesc bd = blocks[it];
        encode_block<AA>(l, enc_tables, seq, bd, lane);

        for (int g = 0; g < NG; g++) {
            uint64_t val[RPG], home[RPG];
            uint32_t fp[RPG];
            bool valid[RPG];
#pragma unroll
            for (int q = 0; q < RPG; q++) {
                uint32_t hi, lo;
                uint64_t quo;
                valid[q] = row_halves<AA>(l, g * RPG + q, lane, bd, &hi, &lo);
                home[q] = split_halves(hi, lo, tab, &quo);
                fp[q] = tag_qs(quo, home[q]);
                val[q] = (uint64_t)hi * 160000ull + lo;
            }

            // Tables whose bit-per-slot digest stays in the L2 (a 20 M-slot table: 2.5 MB of bits against 20 MB of tags, which come
            // from the memory-side cache at a quarter of the L2's gather rate): a query whose home slot is the home of no
            // findable key (61 % at load 0.49) is a miss for certain and never asks for its tags; its walk would have ended
            // with the stream iff the home slot lies in the occupied run at the table's end (as in bucket_index_kernel).
            if (!COUNTERS && hbits) {
                uint32_t w[RPG];
#pragma unroll
                for (int q = 0; q < RPG; q++) w[q] = valid[q] && home[q] < limit ? hbits[home[q] >> 5] : ~0u;
#pragma unroll
                for (int q = 0; q < RPG; q++)
                    if (valid[q] && home[q] < limit && !((w[q] >> ((uint32_t)home[q] & 31u)) & 1u)) {
                        valid[q] = false;
                        if (home[q] >= tail_start) ran_off = true;
                    }
            }
            Payload ent[RPG];
            uint32_t fslot[RPG];
#pragma unroll
            for (int q = 0; q < RPG; q++) fslot[q] = 0;
            const uint32_t foundm = probe_n<RPG, COUNTERS>(tab, val, home, fp, valid, ent, ctr_valid, ctr_slots, ran_off,
                                                           COUNTERS ? prog : nullptr, COUNTERS && stage_slot ? fslot : nullptr);

            // ---- ordered compaction: ballot per row, staging records handed out from the wave's reservation
            uint32_t cnt[RPG], rank[RPG];
            uint32_t total = 0;
#pragma unroll
            for (int q = 0; q < RPG; q++) {
                const int r = g * RPG + q;
                unsigned long long m = __ballot((foundm >> q) & 1u);
                cnt[q] = (uint32_t)__popcll(m);
                // '+' rows ascend with the lane, '-' rows descend: rank so that staging order == position order
                unsigned long long below = m & ((1ull << lane) - 1ull);
                unsigned long long above = lane == 63 ? 0ull : (m >> (lane + 1));
                rank[q] = (!AA && r >= 3) ? (uint32_t)__popcll(above) : (uint32_t)__popcll(below);
                total += cnt[q];
                if (lane == 0) counts[row_index<AA>(bd, it, r)] = cnt[q];
            }
            if (total > res_end - res_at) {                      // uniform: take a new chunk
                unsigned long long chunk = total > stage_chunk ? total : stage_chunk;
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(cursor, chunk);
                b = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                    (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)b);
                res_at = b; res_end = b + chunk;
            }
            const unsigned long long base = res_at;
            res_at += total;
            if (lane == 0) block_stage_base[(uint64_t)it * NG + g] = (uint32_t)base;
            if (total && base + total <= stage_cap) {
                uint32_t rowbase = 0;
#pragma unroll
                for (int q = 0; q < RPG; q++) {
                    if ((foundm >> q) & 1u) {
                        kg_hit h;
                        row_record_key<AA>(bd, g * RPG + q, lane, &h.container, &h.from0InProt);
                        h.oI = ent[q].oI; h.avgOffFromEnd = ent[q].avg; h.fI = ent[q].fI; h.functionWt = ent[q].wt;
                        stage[base + rowbase + rank[q]] = h;
                        if (COUNTERS && stage_slot) stage_slot[base + rowbase + rank[q]] = fslot[q];
                    }
                    rowbase += cnt[q];
                }
            }
        }
        wave_sync();   // LDS is reused by the next block
    }

    if (COUNTERS) flush_counters(ctr_valid, ctr_slots, ctr, lane);
    flush_ran_off(ran_off, ctr, lane);
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
__global__ __launch_bounds__(kWave *kWavesPerWG) void place_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks,
                                                                  const uint32_t *__restrict__ counts,
                                                                  const uint32_t *__restrict__ offs,
                                                                  const uint32_t *__restrict__ block_stage_base, uint32_t rpg,
                                                                  const kg_hit *__restrict__ stage, kg_hit *__restrict__ hits,
                                                                  const uint32_t *__restrict__ stage_slot, uint32_t *__restrict__ hit_slots)
{
    constexpr int ROWS = AA ? 1 : 6;
    const int lane = threadIdx.x & 63;
    const uint32_t it = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerWG + (threadIdx.x >> 6));
    if (it >= n_blocks) return;
    const BlockDesc bd = blocks[it];
    const uint32_t ng = (uint32_t)ROWS / rpg;
    uint32_t src = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if ((uint32_t)r % rpg == 0) src = block_stage_base[(uint64_t)it * ng + (uint32_t)r / rpg];
        const uint32_t vrow = row_index<AA>(bd, it, r);
        uint32_t n = counts[vrow];
        if ((uint32_t)lane < n) {
            hits[(uint64_t)offs[vrow] + lane] = stage[(uint64_t)src + lane];
            if (hit_slots) hit_slots[(uint64_t)offs[vrow] + lane] = stage_slot[(uint64_t)src + lane];      // KG_F_PROGRESS
        }
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
// Several buffers cleared by one launch (a scan starts with seven small clears and one large one: as separate
// memsets they cost ~10 us each, mostly launch gaps).  Sizes in 4-byte words; pointers 4-byte aligned.
struct ClearList { uint32_t *p[8]; uint64_t words[8]; int n; };

__global__ __launch_bounds__(256) void clear_many_kernel(ClearList l)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (int k = 0; k < l.n; k++) {
        uint32_t *p = l.p[k];
        const uint64_t w = l.words[k];
        const uint64_t head = min(w, (uint64_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4u);      // words up to 16-byte alignment
        const uint64_t quads = (w - head) / 4;
        uint4 *q = reinterpret_cast<uint4 *>(p + head);
        for (uint64_t i = tid; i < quads; i += stride) q[i] = make_uint4(0, 0, 0, 0);
        for (uint64_t i = tid; i < head; i += stride) p[i] = 0;
        for (uint64_t i = head + quads * 4 + tid; i < w; i += stride) p[i] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// tag array from the 24-byte records (one pass over the table at load time)
// occupied[0] += occupied slots; occupied[1] = max(1 + index of the last EMPTY slot below limit) = the first slot of the
// occupied run that ends at the end of the record stream (0: no empty slot at all; limit: the last record is empty).
__global__ void build_tags_kernel(const uint8_t *entries, uint64_t limit, uint64_t n_tags, uint64_t num_sigs, uint64_t magic,
                                  uint8_t *tags, unsigned long long *occupied)
{
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic; tab.m35 = 0;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long occ = 0, tail = 0;
    for (; i < n_tags; i += stride) {
        uint32_t t = kTagEmpty;
        if (i < limit) {
            const uint2 *p = reinterpret_cast<const uint2 *>(entries + i * 24);
            uint2 a = p[0];
            int64_t key = (int64_t)(((uint64_t)a.y << 32) | a.x);
            if (key <= KG_MAX_ENCODED) {        // occupied (KGJ:1000); negative keys are occupied and never match
                uint64_t q;                     // same (q, slot) form as the queries; garbage keys get some tag
                const uint64_t slot = split_value((uint64_t)key, tab, &q);
                t = tag_qs(q, slot);
                occ++;
            }
        }
        tags[i] = (uint8_t)t;
        if (t == kTagEmpty && i < limit) tail = i + 1;
    }
    for (int off = 32; off > 0; off >>= 1) {
        occ += __shfl_down(occ, off);
        const unsigned long long o = __shfl_down(tail, off);
        tail = o > tail ? o : tail;
    }
    if ((threadIdx.x & 63) == 0) {
        if (occ) atomicAdd(occupied, occ);
        if (tail) atomicMax(occupied + 1, tail);
    }
}

// ---------------------------------------------------------------------------------------
// Byte home index ("bidx", round 4): ONE byte per slot h, built once per table, that answers "is the k-mer (quotient q,
// home slot h) in the table?" without walking anything.  A query can only match a key whose OWN home slot is h (slot =
// value % numSigs, KGJ:969), and under the reference's lookup (KGJ:944-1034: from the home slot forward, over occupied slots,
// to the k-mer, the first empty slot or the end of the stream; never wrap) such a key is found iff it lies in the occupied
// run that starts at h.  value = q * numSigs + h, so among the keys homed at h the quotient identifies the key.  A bucket of
// the index is as large as a bucket of tags (2 MiB for 2^21 slots: resident in an XCD's L2), so the tag pass can probe it
// instead of the tags: one byte load per query, no fingerprint, no 16-tag window, no walk -- and, for the KmerGuts table,
// no false candidate and no undecided window (the tag pass leaves 65 M candidates per Gbp for 36.7 M hits: fingerprint
// collisions and, above all, home slots inside long occupied runs, whose 16-tag window decides nothing).
// The byte lists the quotient CLASSES c = q % 19 of the findable keys homed at h:
//     0          no such key: a query with home slot h is a miss for certain
//     1..19      one class: c = code - 1
//     20..190    two classes c1 < c2 (171 pairs), bidx_pair_code
//     191..254   three or more classes, hashed to six bits: bit (c % 6) of (code - 191) is set for every listed class; a query
//                whose bit is clear is a miss for certain, one whose bit is set a candidate for the generic walk (1.4 % of
//                the home slots at load 0.5, under half of the queries there)
//     255        the run was not walked to its end (longer than kHomeWalkMax): every query is a candidate for the walk
// With numSigs > 20^8 / 19 (the KmerGuts table: 1.4e9 slots, quotients 0..18) a class IS the quotient and codes 1..190 are
// exact: listed = in the table for certain (kScanOn: the verify pass scans the records from the home slot).  Smaller tables
// fold their quotients into the 19 classes: "not listed" stays a certain miss, "listed" becomes a candidate for the walk --
// same kernels, so that the small tables of the tests and the fuzz run through them.  Negative keys are occupied and match
// nothing; a key stored in front of its home slot or behind a hole is not reachable and not listed -- exactly the
// reference's lookup on hand-made tables too (tests/test_gpu_fullsize.py, hand-made clusters).
// (Round 3's 16-bit home index, probed out of LDS behind a second partition level, was the idea's first form; the second
//  level cost more than it saved and was removed in round 4: profiles/r03_two_level.md.)
constexpr uint32_t kHomeWalkMax = 1024;        // slots walked per home slot before giving up with code 255
constexpr uint32_t kBidxClasses = 19, kBidxPair0 = 20, kBidxHash0 = 191, kBidxMore = 255;
constexpr uint32_t kBidxInexact = 0x80000000u;      // flag in the decoded word: a listed class is a candidate, not a hit

__host__ __device__ constexpr uint32_t bidx_pair_code(uint32_t c1, uint32_t c2)      // c1 < c2 < 19
{
    return kBidxPair0 + c1 * 19u - c1 * (c1 + 1u) / 2u + (c2 - c1 - 1u);
}
static_assert(bidx_pair_code(0, 1) == 20 && bidx_pair_code(0, 18) == 37 && bidx_pair_code(1, 2) == 38 && bidx_pair_code(17, 18) == 190,
              "pair codes must fill 20..190");

// byte -> the 19 class bits (+ kBidxInexact)
__device__ __forceinline__ uint32_t bidx_decode(uint32_t code)
{
    if (code == 0) return 0u;
    if (code < kBidxPair0) return 1u << (code - 1u);
    if (code < kBidxHash0) {
        uint32_t c1 = 0;
        while (code >= bidx_pair_code(c1, c1 + 1u) + (18u - c1)) c1++;
        const uint32_t c2 = c1 + 1u + (code - bidx_pair_code(c1, c1 + 1u));
        return (1u << c1) | (1u << c2);
    }
    if (code == kBidxMore) return 0x7FFFFu | kBidxInexact;
    const uint32_t m6 = code - kBidxHash0;
    uint32_t m = 0;
    for (uint32_t c = 0; c < kBidxClasses; c++) m |= ((m6 >> (c % 6u)) & 1u) << c;
    return m | kBidxInexact;
}

__device__ __forceinline__ uint32_t bidx_encode(uint32_t class_mask, bool complete)
{
    if (!complete) return kBidxMore;
    const uint32_t n = (uint32_t)__popc(class_mask);
    if (n == 0) return 0u;
    const uint32_t c1 = (uint32_t)__builtin_ctz(class_mask);
    if (n == 1) return 1u + c1;
    if (n == 2) return bidx_pair_code(c1, 31u - (uint32_t)__builtin_clz(class_mask));
    const uint32_t m6 = (class_mask | (class_mask >> 6) | (class_mask >> 12) | (class_mask >> 18)) & 63u;
    return kBidxHash0 + m6;
}

__global__ void build_bidx_kernel(const uint8_t *entries, const uint8_t *tags, uint64_t limit, uint64_t n_idx, uint64_t num_sigs,
                                  uint64_t magic, uint8_t *idx)
{
    TableView tab;
    tab.entries = entries; tab.tags = tags; tab.limit = limit; tab.num_sigs = num_sigs; tab.magic = magic; tab.m35 = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n_idx; h += stride) {
        uint32_t mask = 0;
        bool complete = true;
        if (h < limit && tags[h] != kTagEmpty) {
            uint64_t j = h;
            for (; j < limit && j - h < kHomeWalkMax; j++) {
                if (tags[j] == kTagEmpty) break;
                const uint2 a = *reinterpret_cast<const uint2 *>(entries + j * 24);
                const int64_t key = (int64_t)(((uint64_t)a.y << 32) | a.x);
                if (key < 0) continue;                      // occupied, matches no query (KGJ:1000-1004)
                uint64_t q;
                if (split_value((uint64_t)key, tab, &q) != h) continue;
                mask |= 1u << (uint32_t)(q % kBidxClasses);
            }
            if (j < limit && j - h >= kHomeWalkMax) complete = false;
        }
        idx[h] = (uint8_t)bidx_encode(mask, complete);
    }
}

// One BIT per slot of the byte home index: set iff the byte is not 0 (some findable key has its home there, or the run was
// not walked to its end): the direct kernel's first question on tables whose bits fit the L2.  n_words words; bytes behind
// n_idx count as 0.
__global__ void build_hbits_kernel(const uint8_t *__restrict__ idx, uint64_t n_idx, uint32_t *__restrict__ bits, uint64_t n_words)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint32_t out = 0;
        const uint64_t s0 = w * 32;
        if (s0 + 32 <= n_idx) {                                   // (idx is a hipMalloc block: 16-byte aligned at every s0)
            const uint4 a = *reinterpret_cast<const uint4 *>(idx + s0), b = *reinterpret_cast<const uint4 *>(idx + s0 + 16);
            const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int j = 0; j < 4; j++) out |= (((v[k] >> (8 * j)) & 0xFFu) != 0u ? 1u : 0u) << (4 * k + j);
        } else {
            for (uint32_t j = 0; j < 32 && s0 + j < n_idx; j++) out |= (idx[s0 + j] != 0 ? 1u : 0u) << j;
        }
        bits[w] = out;
    }
}

}  // namespace kg
