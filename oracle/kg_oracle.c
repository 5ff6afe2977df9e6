/*
 * kg_oracle.c -- TEST INFRASTRUCTURE ONLY (see kg_oracle.h).
 *
 * Literal, single-threaded CPU restatement of the kmer_guts hot path of
 * rsutormin/KmerGutsJava.  Every function cites the reference lines it follows
 * ("KGJ:n" = lib/src/kmergutsjava/KmerGutsJava.java:n).  Quirks of the
 * reference are reproduced, not fixed: uppercase-only residue codes, the
 * AA-mode missing last window (KGJ:912), linear probing that never wraps
 * (KGJ:799-802 swallow the EOF), the 39 998 hit cap (KGJ:496), sequential
 * float32 weight sums (KGJ:394).  NOT reproduced: the >20 M-k-mer external
 * merge, which drops records (KGJ:705-709); batches are bounded instead
 * (SURVEY.md section 8c "parity definition").
 *
 * PARITY STATUS: parity unpinned (no reference golden vectors exist; no JVM).
 */
#include "kg_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

static char g_err[512];
const char *kgo_last_error(void) { return g_err; }
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* KGJ:111-175  toAminoAcidOff: uppercase ACDEFGHIKLMNPQRSTVWY -> 0..19, else 20 */
int kgo_to_amino_acid_off(int c)
{
    switch (c) {
    case 'A': return 0;  case 'C': return 1;  case 'D': return 2;  case 'E': return 3;
    case 'F': return 4;  case 'G': return 5;  case 'H': return 6;  case 'I': return 7;
    case 'K': return 8;  case 'L': return 9;  case 'M': return 10; case 'N': return 11;
    case 'P': return 12; case 'Q': return 13; case 'R': return 14; case 'S': return 15;
    case 'T': return 16; case 'V': return 17; case 'W': return 18; case 'Y': return 19;
    }
    return 20;
}

/* KGJ:177-260  compl (note the 's' -> 'S' quirk at KGJ:218-219) */
int kgo_compl(int c)
{
    switch (c) {
    case 'a': return 't'; case 'A': return 'T';
    case 'c': return 'g'; case 'C': return 'G';
    case 'g': return 'c'; case 'G': return 'C';
    case 't': case 'u': return 'a';
    case 'T': case 'U': return 'A';
    case 'm': return 'k'; case 'M': return 'K';
    case 'r': return 'y'; case 'R': return 'Y';
    case 'w': return 'w'; case 'W': return 'W';
    case 's': return 'S'; case 'S': return 'S';
    case 'y': return 'r'; case 'Y': return 'R';
    case 'k': return 'm'; case 'K': return 'M';
    case 'b': return 'v'; case 'B': return 'V';
    case 'd': return 'h'; case 'D': return 'H';
    case 'h': return 'd'; case 'H': return 'D';
    case 'v': return 'b'; case 'V': return 'B';
    case 'n': return 'n'; case 'N': return 'N';
    }
    return c;
}

/* KGJ:263-272 */
void kgo_rev_comp(const uint8_t *in, int64_t n, uint8_t *out)
{
    int64_t p = n - 1, pc = 0;
    while (n-- > 0) out[pc++] = (uint8_t)kgo_compl(in[p--]);
}

/* KGJ:274-292 (the "> MAX_ENCODED" throw at KGJ:283 is unreachable: 8 codes < 20) */
int64_t kgo_encoded_kmer(const uint8_t *data, int64_t pos)
{
    int64_t encodedK = 0;
    for (int i = 0; i < KGO_K; i++) {
        int add = data[pos + i];
        if (add >= 20) return -1;
        encodedK = encodedK * 20 + add;
    }
    return encodedK;
}

/* KGJ:294-318 */
int kgo_dna_char(int c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'u': case 'T': case 'U': return 3;
    }
    return 4;
}

/* KGJ:88-93 */
static const char GENETIC_CODE[64] = {
    'K','N','K','N','T','T','T','T','R','S','R','S','I','I','M','I',
    'Q','H','Q','H','P','P','P','P','R','R','R','R','L','L','L','L',
    'E','D','E','D','A','A','A','A','G','G','G','G','V','V','V','V',
    '*','Y','*','Y','S','S','S','S','*','C','W','C','L','F','L','F'
};

/* KGJ:320-343 */
void kgo_translate(const uint8_t *seq, int64_t L, int off, uint8_t *pseq, uint8_t *pIseq, int64_t plen)
{
    int64_t max = L - 3;
    int64_t p = 0;
    for (int64_t i = off; i <= max; ) {
        int c1 = kgo_dna_char(seq[i++]);
        int c2 = kgo_dna_char(seq[i++]);
        int c3 = kgo_dna_char(seq[i++]);
        if (c1 < 4 && c2 < 4 && c3 < 4) {
            int I = c1 * 16 + c2 * 4 + c3;
            char protC = GENETIC_CODE[I];
            pseq[p] = (uint8_t)protC;
            pIseq[p] = (uint8_t)kgo_to_amino_acid_off(protC);
        } else {
            pseq[p] = 'x';
            pIseq[p] = 20;
        }
        p++;
    }
    if (p < plen) {
        pseq[p] = 0;
        pIseq[p] = 21;
    }
}

/* ------------------------------------------------------------------ */
/* growable arrays */
typedef struct { int64_t value; int32_t hitCntId; int32_t protPos; } query_kmer;  /* KGJ:1200-1204 */
typedef struct { query_kmer *a; int64_t n, cap; } qvec;
typedef struct { kgo_hit_rec *a; int64_t n, cap; } hvec;
typedef struct { kgo_call_rec *a; int64_t n, cap; } cvec;

static int qpush(qvec *v, query_kmer q)
{
    if (v->n == v->cap) {
        int64_t nc = v->cap ? v->cap * 2 : 1 << 16;
        query_kmer *na = (query_kmer *)realloc(v->a, (size_t)nc * sizeof *na);
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->n++] = q;
    return 0;
}
static int hpush(hvec *v, kgo_hit_rec h)
{
    if (v->n == v->cap) {
        int64_t nc = v->cap ? v->cap * 2 : 1 << 12;
        kgo_hit_rec *na = (kgo_hit_rec *)realloc(v->a, (size_t)nc * sizeof *na);
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->n++] = h;
    return 0;
}
static int cpush(cvec *v, kgo_call_rec c)
{
    if (v->n == v->cap) {
        int64_t nc = v->cap ? v->cap * 2 : 1 << 10;
        kgo_call_rec *na = (kgo_call_rec *)realloc(v->a, (size_t)nc * sizeof *na);
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->n++] = c;
    return 0;
}

/* ------------------------------------------------------------------ */
/* KGJ:900-922 addKmers: one container, every window i in [0, plen-K) that encodes */
static int add_kmers(const uint8_t *pIseq, int64_t plen, int32_t hitCntId, qvec *q, int64_t *valid)
{
    for (int64_t i = 0; i < plen - KGO_K; i++) {
        int64_t value = kgo_encoded_kmer(pIseq, i);
        if (value < 0) continue;
        query_kmer qk; qk.value = value; qk.protPos = (int32_t)i; qk.hitCntId = hitCntId;
        if (qpush(q, qk)) return -1;
        (*valid)++;
    }
    return 0;
}

/* KGJ:1051-1074 prepareQuery.  Returns number of containers added (1 or 6), <0 on OOM. */
static int prepare_query(const kgo_params *p, const uint8_t *seq, int64_t L, int32_t first_container,
                         qvec *q, int64_t *residues, int64_t *valid)
{
    if (p->aa) {
        uint8_t *pIseq = (uint8_t *)malloc((size_t)(L > 0 ? L : 1));
        if (!pIseq) return -1;
        for (int64_t i = 0; i < L; i++) pIseq[i] = (uint8_t)kgo_to_amino_acid_off(seq[i]);
        int rc = add_kmers(pIseq, L, first_container, q, valid);
        free(pIseq);
        *residues += L;
        return rc ? -1 : 1;
    }
    int64_t len = L / 3 + 1;                      /* KGJ:1061 */
    uint8_t *pseq = (uint8_t *)calloc((size_t)len, 1);
    uint8_t *pIseq = (uint8_t *)calloc((size_t)len, 1);   /* Java arrays start zeroed; reused for all 6 frames */
    uint8_t *rc = (uint8_t *)malloc((size_t)(L > 0 ? L : 1));
    if (!pseq || !pIseq || !rc) { free(pseq); free(pIseq); free(rc); return -1; }
    int err = 0;
    for (int frame = 0; frame < 3 && !err; frame++) {
        kgo_translate(seq, L, frame, pseq, pIseq, len);
        err = add_kmers(pIseq, len, first_container + frame, q, valid);
        if (L - frame >= 3) *residues += (L - frame) / 3;
    }
    kgo_rev_comp(seq, L, rc);
    for (int frame = 0; frame < 3 && !err; frame++) {
        kgo_translate(rc, L, frame, pseq, pIseq, len);
        err = add_kmers(pIseq, len, first_container + 3 + frame, q, valid);
        if (L - frame >= 3) *residues += (L - frame) / 3;
    }
    free(pseq); free(pIseq); free(rc);
    return err ? -1 : 6;
}

/* KGJ:1082-1095 comparator (value % numSigs, value); KGJ:1079 Collections.sort is a stable merge sort */
static int64_t g_numSigs;
static inline int qcmp(const query_kmer *o1, const query_kmer *o2)
{
    int64_t h1 = o1->value % g_numSigs;
    int64_t h2 = o2->value % g_numSigs;
    if (h1 != h2) return h1 < h2 ? -1 : 1;
    if (o1->value != o2->value) return o1->value < o2->value ? -1 : 1;
    return 0;
}
static void msort(query_kmer *a, query_kmer *tmp, int64_t n)
{
    if (n < 2) return;
    if (n <= 12) {   /* insertion sort, stable */
        for (int64_t i = 1; i < n; i++) {
            query_kmer x = a[i];
            int64_t j = i;
            while (j > 0 && qcmp(&a[j - 1], &x) > 0) { a[j] = a[j - 1]; j--; }
            a[j] = x;
        }
        return;
    }
    int64_t h = n / 2;
    msort(a, tmp, h);
    msort(a + h, tmp, n - h);
    if (qcmp(&a[h - 1], &a[h]) <= 0) return;
    memcpy(tmp, a, (size_t)h * sizeof *a);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (qcmp(&a[j], &tmp[i]) < 0) a[k++] = a[j++];
        else a[k++] = tmp[i++];
    }
    while (i < h) a[k++] = tmp[i++];
}

/* KGJ:1097-1130 little-endian readers over the in-memory file image */
static inline int64_t rd_i64le(const uint8_t *b)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    return (int64_t)v;
}
static inline int32_t rd_i32le(const uint8_t *b)
{
    uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    return (int32_t)v;
}
static inline float rd_f32le(const uint8_t *b)
{
    int32_t i = rd_i32le(b); float f; memcpy(&f, &i, 4); return f;
}

/* KGJ:944-1034 lookup, literal: stream the table once, merge-join with the sorted queries.
 * "file" is the stream after the 24-byte header.  Returns 1 if the stream ran off the end
 * (EOFException / failed skip -> swallowed at KGJ:799-802), 0 otherwise, -1 on OOM. */
typedef struct { int64_t value; int64_t first, count; } group;   /* one inProgress map entry */
static int lookup_literal(const uint8_t *file, int64_t file_n, int64_t numSigs, int64_t entrySize,
                          const query_kmer *qs, int64_t nq, hvec *hits, kgo_result *out)
{
    int64_t cursor = 0;                 /* stream position */
    int64_t curHashCode = 0;
    int64_t cur = 0;                    /* kmerStorage.loadNext() index */
    group *inProgress = NULL; int64_t ng = 0, gcap = 0;
    int aborted = 0;
    int64_t kmersFound = 0;             /* KGJ:956 */
    int fraction = 0;                   /* KGJ:963 */
    while (cur < nq || ng > 0) {
        int64_t neededHashCode = curHashCode;
        if (ng == 0) {                                          /* KGJ:966-974 */
            neededHashCode = qs[cur].value % numSigs;
            if (gcap == 0) { gcap = 16; inProgress = (group *)malloc((size_t)gcap * sizeof *inProgress); if (!inProgress) return -1; }
            inProgress[0].value = qs[cur].value; inProgress[0].first = cur; inProgress[0].count = 1;
            ng = 1; cur++;
        }
        while (cur < nq) {                                      /* KGJ:976-989 */
            if (qs[cur].value % numSigs != neededHashCode) break;
            int64_t g;
            for (g = 0; g < ng; g++) if (inProgress[g].value == qs[cur].value) break;
            if (g < ng) {
                /* sorted by (hash, value): members of one group are contiguous */
                inProgress[g].count++;
            } else {
                if (ng == gcap) {
                    gcap *= 2;
                    group *na = (group *)realloc(inProgress, (size_t)gcap * sizeof *na);
                    if (!na) { free(inProgress); return -1; }
                    inProgress = na;
                }
                inProgress[ng].value = qs[cur].value; inProgress[ng].first = cur; inProgress[ng].count = 1;
                ng++;
            }
            cur++;
        }
        if (neededHashCode > curHashCode) {                     /* KGJ:991-994 skipBytesFully */
            int64_t skip = entrySize * (neededHashCode - curHashCode);
            if (skip > file_n - cursor) {                       /* KGJ:1045-1047 -> swallowed */
                if (out->skip_failed_bytes < 0 && !out->read_eof) out->skip_failed_bytes = skip;
                aborted = 1; break;
            }
            if (skip > 0) cursor += skip;
            curHashCode = neededHashCode;
        }
        if (file_n - cursor < 24) {                             /* EOFException KGJ:1102,1116 */
            if (out->skip_failed_bytes < 0) out->read_eof = 1;
            aborted = 1; break;
        }
        const uint8_t *e = file + cursor;                       /* KGJ:995-999 */
        int64_t whichKmer = rd_i64le(e);
        int32_t otuIndex = rd_i32le(e + 8);
        int32_t avgFromEnd = rd_i32le(e + 12);
        int32_t functionIndex = rd_i32le(e + 16);
        float functionWt = rd_f32le(e + 20);
        cursor += 24;
        if (whichKmer > KGO_MAX_ENCODED) {                      /* KGJ:1000-1001 */
            ng = 0;
        } else {
            int64_t g;
            for (g = 0; g < ng; g++) if (inProgress[g].value == whichKmer) break;
            if (g < ng) {                                       /* KGJ:1004-1015 */
                kmersFound++;
                for (int64_t k = 0; k < inProgress[g].count; k++) {
                    const query_kmer *qk = &qs[inProgress[g].first + k];
                    kgo_hit_rec h;
                    h.container = (uint32_t)qk->hitCntId;
                    h.from0InProt = qk->protPos;
                    h.oI = otuIndex; h.avgOffFromEnd = avgFromEnd; h.fI = functionIndex; h.functionWt = functionWt;
                    if (hpush(hits, h)) { free(inProgress); return -1; }
                }
                inProgress[g] = inProgress[ng - 1];
                ng--;
            }
        }
        curHashCode++;
        {                                                       /* KGJ:1017-1024 */
            int newFraction = (int)(10.0 * ((double)curHashCode / (double)numSigs));
            if (newFraction != fraction) {
                fraction = newFraction;
                if (out->n_processed < KGO_MAX_PROCESSED) {
                    out->processed_tenth[out->n_processed] = fraction;
                    out->processed_found[out->n_processed] = kmersFound;
                    out->n_processed++;
                }
            }
        }
    }
    out->kmers_found += kmersFound;
    free(inProgress);
    return aborted;
}

/* Independent linear probing without wrap-around: the semantics the literal merge-join
 * implements.  Counts every table entry inspected. */
static int lookup_direct(const uint8_t *file, int64_t file_n, int64_t numSigs,
                         const query_kmer *qs, int64_t nq, hvec *hits, int64_t *inspected, int32_t *ran_off)
{
    /* the reference never compares the running slot with numSigs (KGJ:964-1026): a probe walk ends
     * at an empty slot, at the k-mer, or where the stream ends (EOF == not found) */
    int64_t limit = file_n / 24;
    for (int64_t k = 0; k < nq; k++) {
        int64_t v = qs[k].value;
        int64_t s;
        for (s = v % numSigs; s < limit; s++) {
            const uint8_t *e = file + s * 24;
            int64_t whichKmer = rd_i64le(e);
            (*inspected)++;
            if (whichKmer > KGO_MAX_ENCODED) break;
            if (whichKmer == v) {
                kgo_hit_rec h;
                h.container = (uint32_t)qs[k].hitCntId;
                h.from0InProt = qs[k].protPos;
                h.oI = rd_i32le(e + 8); h.avgOffFromEnd = rd_i32le(e + 12);
                h.fI = rd_i32le(e + 16); h.functionWt = rd_f32le(e + 20);
                if (hpush(hits, h)) return -1;
                break;
            }
        }
        /* undecided at the end of the stream: where the reference's merge-join dies with EOFException (or cannot
         * skip to the home slot of a truncated file), KGJ:799-802 */
        if (s >= limit) *ran_off = 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* aggregation */
typedef struct { kgo_hit_rec *a; int64_t n, cap; } hitlist;     /* the Java "hits" ArrayList */

/* KGJ:385-455 processSetOfHits.  Returns new currentFI; *crash set on the reference's exception path. */
static int process_set_of_hits(const kgo_params *p, hitlist *hits, int currentFI, kgo_otu_rec *oi,
                               uint32_t container, cvec *calls, int *crash)
{
    int fICount = 0;
    float weightedHits = 0;
    int64_t lastHit = 0;
    for (int64_t i = 0; i < hits->n; i++) {                     /* KGJ:390-396 */
        if (hits->a[i].fI == currentFI) {
            lastHit = i;
            fICount++;
            weightedHits += hits->a[i].functionWt;             /* float32, list order */
        }
    }
    if (fICount >= p->min_hits && weightedHits >= (float)p->min_weighted_hits) {   /* KGJ:397 */
        if (hits->n == 0) { *crash = 1; return currentFI; }     /* hits.get(0) would throw */
        kgo_call_rec c;
        c.container = container;
        c.start = hits->a[0].from0InProt;                       /* KGJ:399 */
        c.end = hits->a[lastHit].from0InProt + (KGO_K - 1);     /* KGJ:400 */
        c.count = fICount; c.fI = currentFI; c.weightedHits = weightedHits;
        if (cpush(calls, c)) { *crash = 2; return currentFI; }
        for (int64_t i = 0; i <= lastHit; i++) {                /* KGJ:413-439 */
            if (hits->a[i].fI == currentFI) {
                int j;
                for (j = 0; j < oi->n && oi->oI[j] != hits->a[i].oI; j++) {}
                if (j == oi->n) {
                    if (oi->n == KGO_OI_BUFSZ) j--;             /* overwrite the last entry */
                    else oi->n++;
                    oi->oI[j] = hits->a[i].oI;
                    oi->count[j] = 1;
                } else {
                    oi->count[j]++;
                }
                while (j > 0 && oi->count[j - 1] <= oi->count[j]) {   /* KGJ:432-437 */
                    int32_t tc = oi->count[j - 1], to = oi->oI[j - 1];
                    oi->count[j - 1] = oi->count[j]; oi->oI[j - 1] = oi->oI[j];
                    oi->count[j] = tc; oi->oI[j] = to;
                    j--;
                }
            }
        }
    }
    int64_t numHits = hits->n;                                  /* KGJ:441-453 */
    if (numHits < 2) { *crash = 1; return currentFI; }          /* hits.get(numHits-2) throws */
    if (hits->a[numHits - 2].fI != currentFI && hits->a[numHits - 2].fI == hits->a[numHits - 1].fI) {
        currentFI = hits->a[numHits - 1].fI;
        hits->a[0] = hits->a[numHits - 2];
        hits->a[1] = hits->a[numHits - 1];
        hits->n = 2;
    } else {
        hits->n = 0;
    }
    return currentFI;
}

static int hit_pos_cmp(const void *a, const void *b)
{
    const kgo_hit_rec *x = (const kgo_hit_rec *)a, *y = (const kgo_hit_rec *)b;
    if (x->container != y->container) return x->container < y->container ? -1 : 1;
    if (x->from0InProt != y->from0InProt) return x->from0InProt < y->from0InProt ? -1 : 1;
    return 0;
}

/* KGJ:457-514 gatherHits on hits already sorted by from0InProt.
 * trace (n bytes, may be NULL) / *tail: what the -d stream of the reference would show at each record, in the
 * KGO_EV_* encoding: whether "after-hit" is printed (the record joined the list), whether the list was processed
 * or cleared before / after that, whether that printed CALL + "after-call", and whether the list then kept its
 * last two members (KGJ:406-409, 441-453, 470-473, 498-501). */
static int gather_sorted(const kgo_params *p, const kgo_hit_rec *all, int64_t n, uint32_t container,
                         kgo_otu_rec *oi, cvec *calls, hitlist *hits, uint8_t *trace, uint8_t *tail)
{
    int crash = 0;
    hits->n = 0;
    int currentFI = 0;
    if (tail) *tail = 0;
    for (int64_t k = 0; k < n && !crash; k++) {
        const kgo_hit_rec *ph = &all[k];
        int avgOffEnd = ph->avgOffFromEnd;
        int fI = ph->fI;
        unsigned ev = 0;
        if (hits->n > 0 &&
            (int32_t)((uint32_t)hits->a[hits->n - 1].from0InProt + (uint32_t)p->max_gap) < ph->from0InProt) {   /* KGJ:477-484 */
            ev |= KGO_EV_RESET_BEFORE;
            if (hits->n >= p->min_hits) {
                int64_t before = calls->n;
                currentFI = process_set_of_hits(p, hits, currentFI, oi, container, calls, &crash);
                if (calls->n > before) ev |= KGO_EV_CALL_BEFORE;
                if (hits->n == 2) ev |= KGO_EV_KEEP2_BEFORE;
            } else
                hits->n = 0;
            if (crash) break;
        }
        if (hits->n == 0) currentFI = fI;                       /* KGJ:486-488 */
        int accept = !p->order_constraint || hits->n == 0;
        if (!accept) {                                          /* KGJ:490-494 */
            const kgo_hit_rec *last = &hits->a[hits->n - 1];
            /* Java int arithmetic wraps; signed overflow is undefined in C (found by the UBSan build, K17): unsigned */
            int32_t d = (int32_t)(((uint32_t)ph->from0InProt - (uint32_t)last->from0InProt) -
                                  ((uint32_t)last->avgOffFromEnd - (uint32_t)avgOffEnd));
            /* Math.abs(int): abs(MIN_VALUE) stays negative in Java */
            int32_t ad = d < 0 ? (int32_t)(0u - (uint32_t)d) : d;
            accept = (fI == last->fI) && (ad <= 20);
        }
        if (accept) {
            if (hits->n < KGO_MAX_HITS_PER_SEQ - 2) {           /* KGJ:496-497 */
                if (hits->n == hits->cap) {
                    int64_t nc = hits->cap ? hits->cap * 2 : 64;
                    kgo_hit_rec *na = (kgo_hit_rec *)realloc(hits->a, (size_t)nc * sizeof *na);
                    if (!na) return -2;
                    hits->a = na; hits->cap = nc;
                }
                hits->a[hits->n++] = *ph;
                ev |= KGO_EV_ACCEPTED;
            }
            if (hits->n > 1 && currentFI != fI &&
                hits->a[hits->n - 2].fI == hits->a[hits->n - 1].fI) {      /* KGJ:503-508 */
                int64_t before = calls->n;
                ev |= KGO_EV_RESET_AFTER;
                currentFI = process_set_of_hits(p, hits, currentFI, oi, container, calls, &crash);
                if (calls->n > before) ev |= KGO_EV_CALL_AFTER;
                if (hits->n == 2) ev |= KGO_EV_KEEP2_AFTER;
            }
        }
        if (trace) trace[k] = (uint8_t)ev;
    }
    if (!crash && hits->n >= p->min_hits) {                      /* KGJ:511-513 */
        int64_t before = calls->n;
        process_set_of_hits(p, hits, currentFI, oi, container, calls, &crash);
        if (tail && calls->n > before) *tail = KGO_EV_TAIL_CALL;
    }
    return crash ? -crash : 0;
}

int64_t kgo_gather_hits(const kgo_params *p, kgo_hit_rec *hits, int64_t n, uint32_t container,
                        kgo_otu_rec *otu, kgo_call_rec *calls_out, int64_t cap)
{
    for (int64_t i = 0; i < n; i++) hits[i].container = container;
    /* KGJ:460-465 stable sort by from0InProt: merge sort via qsort on (pos, original index) */
    typedef struct { kgo_hit_rec h; int64_t idx; } tagged;
    tagged *t = (tagged *)malloc((size_t)(n > 0 ? n : 1) * sizeof *t);
    if (!t) return -1;
    for (int64_t i = 0; i < n; i++) { t[i].h = hits[i]; t[i].idx = i; }
    /* simple stable insertion/merge: use bottom-up merge sort */
    tagged *tmp = (tagged *)malloc((size_t)(n > 0 ? n : 1) * sizeof *tmp);
    if (!tmp) { free(t); return -1; }
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (t[j].h.from0InProt < t[i].h.from0InProt) tmp[k++] = t[j++];
                else tmp[k++] = t[i++];
            }
            while (i < mid) tmp[k++] = t[i++];
            while (j < hi) tmp[k++] = t[j++];
        }
        tagged *s = t; t = tmp; tmp = s;
    }
    for (int64_t i = 0; i < n; i++) hits[i] = t[i].h;
    free(t); free(tmp);
    cvec calls = {0};
    hitlist hl = {0};
    int rc = gather_sorted(p, hits, n, container, otu, &calls, &hl, NULL, NULL);
    free(hl.a);
    int64_t nc = calls.n;
    for (int64_t i = 0; i < nc && i < cap; i++) calls_out[i] = calls.a[i];
    free(calls.a);
    if (rc) return -1;
    return nc;
}

/* ------------------------------------------------------------------ */
void kgo_result_free(kgo_result *r)
{
    if (!r) return;
    free(r->hits); free(r->container_hit_start); free(r->calls);
    free(r->container_call_start); free(r->otu); free(r->hit_events); free(r->container_tail_events);
    memset(r, 0, sizeof *r);
}

/* KGJ:742-820 run(), minus file handling and text. */
int kgo_run(const uint8_t *table, size_t table_nbytes, const kgo_params *p,
            const uint8_t *seq, const int64_t *off, int64_t n_seqs, int lookup_mode,
            kgo_result *out)
{
    memset(out, 0, sizeof *out);
    out->skip_failed_bytes = -1;
    if (table_nbytes < 24) return fail("table image shorter than its 24-byte header");
    /* KGJ:933-935 */
    int64_t numSigs = rd_i64le(table);
    int64_t entrySize = rd_i64le(table + 8);
    /* int64_t version = rd_i64le(table + 16);  never checked by the reference */
    if (numSigs <= 0) return fail("numSigs <= 0 (value % numSigs would throw in the reference)");
    if (p->min_hits < 2) return fail("minHits < 2 makes the reference throw in processSetOfHits (KGJ:442)");
    const uint8_t *file = table + 24;
    int64_t file_n = (int64_t)table_nbytes - 24;
    g_numSigs = numSigs;

    int per = p->aa ? 1 : 6;
    int64_t n_cont = n_seqs * per;
    hvec hits = {0};
    qvec q = {0};
    query_kmer *tmp = NULL; int64_t tmpcap = 0;
    int64_t limit = p->input_size_limit > 0 ? p->input_size_limit : 20000000;
    int rc = 0;

    int64_t s = 0;
    while (s < n_seqs && rc == 0) {
        /* one batch: as many whole sequences as stay within inputSizeLimit query k-mers (KGJ:108, 832) */
        double t0 = now_s();
        q.n = 0;
        int64_t s0 = s;
        while (s < n_seqs) {
            int64_t L = off[s + 1] - off[s];
            int64_t est = p->aa ? L : 2 * L;
            if (s > s0 && q.n + est > limit) break;
            int nc = prepare_query(p, seq + off[s], L, (int32_t)(s * per), &q, &out->residues, &out->windows_valid);
            if (nc < 0) { rc = fail("out of memory in prepareQuery"); break; }
            s++;
        }
        if (rc) break;
        if (lookup_mode == 0) {
            if (q.n > tmpcap) {
                free(tmp); tmpcap = q.n;
                tmp = (query_kmer *)malloc((size_t)(tmpcap / 2 + 1) * sizeof *tmp);
                if (!tmp) { rc = fail("out of memory in sort"); break; }
            }
            msort(q.a, tmp, q.n);                                /* KGJ:847 finalizeSorting */
        }
        double t1 = now_s();
        out->t_prepare += t1 - t0;
        if (lookup_mode == 0) {
            int a = lookup_literal(file, file_n, numSigs, entrySize, q.a, q.n, &hits, out);
            if (a < 0) { rc = fail("out of memory in lookup"); break; }
            if (a) out->lookup_aborted = 1;
        } else {
            if (entrySize != 24) { rc = fail("direct-probe mode needs entrySize == 24"); break; }
            if (lookup_direct(file, file_n, numSigs, q.a, q.n, &hits, &out->slots_inspected, &out->lookup_aborted)) {
                rc = fail("out of memory in lookup"); break;
            }
        }
        out->t_lookup += now_s() - t1;
    }
    free(q.a); free(tmp);
    if (rc) { free(hits.a); return rc; }

    /* KGJ:805-818 grouping: per query id in FASTA order; per container gatherHits */
    double t3 = now_s();
    /* gatherHits' stable sort by from0InProt (KGJ:460-465); positions are unique per container */
    if (hits.n) qsort(hits.a, (size_t)hits.n, sizeof *hits.a, hit_pos_cmp);      /* (no hit at all: hits.a is NULL) */
    out->n_seqs = n_seqs; out->n_containers = n_cont;
    out->container_hit_start = (int64_t *)calloc((size_t)n_cont + 1, sizeof(int64_t));
    out->container_call_start = (int64_t *)calloc((size_t)n_cont + 1, sizeof(int64_t));
    out->otu = (kgo_otu_rec *)calloc((size_t)(n_seqs > 0 ? n_seqs : 1), sizeof(kgo_otu_rec));
    out->hit_events = (uint8_t *)calloc((size_t)(hits.n > 0 ? hits.n : 1), 1);
    out->container_tail_events = (uint8_t *)calloc((size_t)(n_cont > 0 ? n_cont : 1), 1);
    if (!out->container_hit_start || !out->container_call_start || !out->otu || !out->hit_events ||
        !out->container_tail_events) {
        free(hits.a); kgo_result_free(out); return fail("out of memory in grouping");
    }
    {
        int64_t k = 0;
        for (int64_t c = 0; c < n_cont; c++) {
            out->container_hit_start[c] = k;
            while (k < hits.n && hits.a[k].container == (uint32_t)c) k++;
        }
        out->container_hit_start[n_cont] = k;
    }
    cvec calls = {0};
    hitlist hl = {0};
    for (int64_t sq = 0; sq < n_seqs && rc == 0; sq++) {
        kgo_otu_rec *oi = &out->otu[sq];                         /* KGJ:528,540: one buffer per sequence */
        for (int f = 0; f < per; f++) {                          /* KGJ:542-555: + 0,1,2 then - 0,1,2 */
            int64_t c = sq * per + f;
            out->container_call_start[c] = calls.n;
            int64_t a = out->container_hit_start[c], b = out->container_hit_start[c + 1];
            int g = gather_sorted(p, hits.a + a, b - a, (uint32_t)c, oi, &calls, &hl, out->hit_events + a,
                                  out->container_tail_events + c);
            if (g) { rc = fail("reference crash path reached in processSetOfHits / out of memory"); break; }
        }
    }
    free(hl.a);
    if (rc) { free(hits.a); free(calls.a); kgo_result_free(out); return rc; }
    out->container_call_start[n_cont] = calls.n;
    out->hits = hits.a; out->n_hits = hits.n;
    out->calls = calls.a; out->n_calls = calls.n;
    out->t_group = now_s() - t3;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Java Formatter %.<p>f of (double)(float): decimal digits of the value, rounded HALF_UP.
 * For a float the only inputs on which HALF_UP and C's exact-binary round-half-even differ are
 * exact ties, i.e. v * 2^(p+1) an odd integer (SURVEY 8c note N3). */
int kgo_format_java_f(float v, int precision, char *buf, size_t bufsz)
{
    double d = (double)v;
    if (isnan(d)) return snprintf(buf, bufsz, "NaN");
    if (isinf(d)) return snprintf(buf, bufsz, d < 0 ? "-Infinity" : "Infinity");
    double ad = fabs(d);
    double scaled = ldexp(ad, precision + 1);      /* exact */
    if (precision >= 0 && precision <= 9 && scaled < 9.0e15 && scaled == floor(scaled) &&
        fmod(scaled, 2.0) == 1.0) {
        /* tie: ad = k / 2^(p+1), k odd.  ad * 10^p = k * 5^p / 2 = (k*5^p - 1)/2 + 0.5 -> HALF_UP */
        uint64_t k = (uint64_t)scaled;
        uint64_t p5 = 1, p10 = 1;
        for (int i = 0; i < precision; i++) { p5 *= 5; p10 *= 10; }
        if (k < UINT64_MAX / p5 - 1) {
            uint64_t units = (k * p5 + 1) / 2;
            uint64_t ip = units / p10, fp = units % p10;
            if (precision == 0) return snprintf(buf, bufsz, "%s%llu", signbit(d) ? "-" : "", (unsigned long long)ip);
            return snprintf(buf, bufsz, "%s%llu.%0*llu", signbit(d) ? "-" : "", (unsigned long long)ip,
                            precision, (unsigned long long)fp);
        }
    }
    return snprintf(buf, bufsz, "%.*f", precision, d);
}
