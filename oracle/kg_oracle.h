/*
 * kg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the kmer_guts hot path of the reference
 * (rsutormin/KmerGutsJava).  "KGJ:n" below means
 * /root/reference/lib/src/kmergutsjava/KmerGutsJava.java line n.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (kmergutsjava_amd/, libkmerguts_hip.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference holds no golden vector for
 * this path (its single JUnit test asserts nothing and its k-mer table is not
 * in the repository) and no JVM exists in the build
 * image, so this restatement is pinned only by the hand-derived known-answer
 * tests K1..K15 (tests/test_oracle_kat.py) and by agreement with a second,
 * independently written pure-Python model (oracle/kgj_model.py).
 */
#ifndef KG_ORACLE_H
#define KG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KGJ:85-99 */
#define KGO_K 8
#define KGO_CORE 1280000000LL            /* 20^7 */
#define KGO_MAX_ENCODED 25600000000LL    /* 20^8 */
#define KGO_MAX_HITS_PER_SEQ 40000
#define KGO_OI_BUFSZ 5

/* KGJ:102-108 -- the instance fields the hot path reads */
typedef struct {
    int32_t aa;                /* -a */
    int32_t order_constraint;  /* -O */
    int32_t min_hits;          /* -m, default 5 */
    int32_t min_weighted_hits; /* -M, default 0 */
    int32_t max_gap;           /* -g, default 200 */
    int32_t reserved;
    int64_t input_size_limit;  /* KGJ:108, 20 000 000; batching bound of the literal lookup */
} kgo_params;

/* binary records, same layout as include/kmerguts_hip.h */
typedef struct { uint32_t container; int32_t from0InProt, oI, avgOffFromEnd, fI; float functionWt; } kgo_hit_rec;
typedef struct { uint32_t container; int32_t start, end, count, fI; float weightedHits; } kgo_call_rec;
typedef struct { int32_t n; int32_t count[KGO_OI_BUFSZ]; int32_t oI[KGO_OI_BUFSZ]; } kgo_otu_rec;

#define KGO_MAX_PROCESSED 64
typedef struct {
    int64_t n_seqs;
    int64_t n_containers;
    int64_t n_hits, n_calls;
    kgo_hit_rec  *hits;             /* ordered by (container, from0InProt)           */
    int64_t      *container_hit_start; /* n_containers+1                              */
    kgo_call_rec *calls;            /* emission order (container, then run order)    */
    int64_t      *container_call_start; /* n_containers+1                             */
    kgo_otu_rec  *otu;              /* one per sequence                              */
    /* counters, SURVEY section 8d */
    int64_t residues;               /* translated positions (DNA) / characters (AA)  */
    int64_t windows_valid;          /* query k-mers (KGJ:913-920)                    */
    int64_t slots_inspected;        /* table entries inspected, direct-probe count   */
    /* phase timers, KGJ:794,803,819 (seconds) */
    double t_prepare, t_lookup, t_group;
    int32_t lookup_aborted;         /* the stream ran off the end (KGJ:799-802) in >=1 batch: literal mode = the merge-join
                                       threw; direct mode = a probe walk reached the end of the records undecided */
    uint8_t *hit_events;            /* n_hits: KGO_EV_* of each record (the -d stream, see gather_sorted) */
    uint8_t *container_tail_events; /* n_containers: KGO_EV_TAIL_CALL                               */
    /* literal mode only: what the reference's table stream reports while the merge-join runs (KGJ:1016-1025: one
     * "Processed: <10 tenth>%, time=..., found-so-far=<kmersFound>" line whenever the tenth of the table changes), in print
     * order over all batches (at most KGO_MAX_PROCESSED lines kept), the distinct k-mers found (KGJ:1004-1006) and how the
     * stream failed, if it did */
    int32_t n_processed;
    int32_t processed_tenth[KGO_MAX_PROCESSED];
    int64_t processed_found[KGO_MAX_PROCESSED];
    int64_t kmers_found;            /* kmersFound summed over the batches                                        */
    int64_t skip_failed_bytes;      /* -1, or N of "Error skipping N bytes" (KGJ:1036-1049): the skip to the next home slot
                                       asked for more than the stream holds.  A .gz stream fails there; a plain file's skip
                                       succeeds and the read behind it throws EOFException instead                        */
    int32_t read_eof;               /* the merge-join read past the end of the stream (EOFException, KGJ:1097-1126)        */
} kgo_result;

/* what the reference's -d stream shows at one hit record, in the order it happens */
#define KGO_EV_ACCEPTED      0x01u  /* "after-hit" printed: the record joined the hits list (KGJ:496-501)     */
#define KGO_EV_RESET_BEFORE  0x02u  /* gap rule fired before it (KGJ:477-484)                                 */
#define KGO_EV_CALL_BEFORE   0x04u  /*   and printed CALL + "after-call" (KGJ:397-409)                        */
#define KGO_EV_KEEP2_BEFORE  0x08u  /*   and the list kept its last two members (KGJ:441-449)                 */
#define KGO_EV_RESET_AFTER   0x10u  /* pair rule fired after it (KGJ:503-508)                                 */
#define KGO_EV_CALL_AFTER    0x20u
#define KGO_EV_KEEP2_AFTER   0x40u
#define KGO_EV_TAIL_CALL     0x01u  /* per container: the final flush printed a CALL (KGJ:511-513)            */

/* ---- single functions (KAT surface) ---- */
int     kgo_to_amino_acid_off(int c);                         /* KGJ:111-175 */
int     kgo_compl(int c);                                     /* KGJ:177-260 */
void    kgo_rev_comp(const uint8_t *in, int64_t n, uint8_t *out);  /* KGJ:263-272 */
int64_t kgo_encoded_kmer(const uint8_t *codes, int64_t pos);  /* KGJ:274-292 */
int     kgo_dna_char(int c);                                  /* KGJ:294-318 */
/* KGJ:320-343; pseq/pIseq have plen entries */
void    kgo_translate(const uint8_t *seq, int64_t L, int off, uint8_t *pseq, uint8_t *pIseq, int64_t plen);

/* gatherHits on one container's hits (KGJ:457-514) + processSetOfHits (KGJ:385-455).
 * hits: n records (any order; sorted stably by from0InProt inside).
 * otu: in/out per-contig buffer.  calls_out: capacity cap; returns number of calls
 * (may exceed cap -> only cap written).  returns -1 on the reference's crash paths. */
int64_t kgo_gather_hits(const kgo_params *p, kgo_hit_rec *hits, int64_t n, uint32_t container,
                        kgo_otu_rec *otu, kgo_call_rec *calls_out, int64_t cap);

/* Whole path: prepareQuery/addKmers -> sort -> lookup -> gatherHits for every sequence.
 * table: image of kmer.table.mem_map (24-byte header + entries, uncompressed).
 * seq/off: concatenated raw sequence characters, off[n_seqs+1].
 * lookup_mode 0 = literal sorted merge-join (KGJ:944-1034, 1076-1095)
 *             1 = independent linear probing without wrap (equivalent; counts slots_inspected)
 * returns 0 ok, <0 error (message via kgo_last_error). */
int kgo_run(const uint8_t *table, size_t table_nbytes, const kgo_params *p,
            const uint8_t *seq, const int64_t *off, int64_t n_seqs, int lookup_mode,
            kgo_result *out);
void kgo_result_free(kgo_result *r);
const char *kgo_last_error(void);

/* Java String.format("%f"/"%1.3f") of a float (promoted to double): HALF_UP on the
 * decimal digits (SURVEY 8c note N3).  returns chars written (excluding NUL). */
int kgo_format_java_f(float v, int precision, char *buf, size_t bufsz);

#ifdef __cplusplus
}
#endif
#endif
