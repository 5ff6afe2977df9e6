/*
 * kmerguts_hip.h -- C ABI of libkmerguts_hip.so, the MI355X (gfx950) implementation of the
 * kmer_guts hot path of rsutormin/KmerGutsJava.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own (it is one Java class);
 * every entry point below names the reference interface it replaces.  "KGJ:n" =
 * lib/src/kmergutsjava/KmerGutsJava.java line n of the reference.  The Java (JNA) and Python
 * (ctypes) bindings that call these are shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns an int status
 * (KG_OK == 0, negative == error) and never throws or aborts across the boundary; the text
 * of the last error on the calling thread is kg_last_error().  All memory returned by the
 * library is owned by the library and released by kg_result_free / kg_table_close.
 * A kg_table is read-only after creation and may be shared by host threads; at most one
 * kg_scan* may be in flight per kg_table at a time (the reference's instance is not
 * re-entrant either, KGJ:838): a second concurrent kg_scan* on the same table returns
 * KG_ERR_BUSY without touching anything.
 *
 * Environment variables read by kg_scan* (tuning and test hooks, none needed in production; every
 * one is read at the start of each call, so a test can change them between calls):
 *   KG_PARTITION (0 direct / 1 partitioned whenever possible / 2 auto), KG_BIDX (0: probe the tags, not the byte home index),
 *   KG_PART_SHIFT, KG_PART_CHUNKS, KG_PART_MIN_CHUNK_BLOCKS, KG_PART_WGS, KG_PART_SLACK, KG_PART_OVF_GROUPS, KG_PART_TAPER,
 *   KG_PROBE_GRID, KG_INDEX_GRID, KG_INDEX_R, KG_PROBE_GRAB, KG_VERIFY_GRID, KG_LOWC_GRID, KG_OVF_GRID,
 *   KG_ORDER_GRID, KG_ORDER_STREAMS, KG_PLACE_STAGED, KG_SCAN_GRID, KG_SCAN_RPG, KG_DIRECT_FILTER, KG_SCATTER_PRIO, KG_INDEX_PRIO, KG_VERIFY_PRIO, KG_STAGE_CHUNK, KG_AGG_PIECES, KG_AGG_PAIRS, KG_AGG_BLOCK_SHIFT:
 *   geometry of the
 *   scan strategies (kmerguts_hip.hip, scan_impl); results never depend on them.  KG_DEBUG: one stderr line per attempt.
 *   TEST HOOKS (used by tests/ only; inert unless the process set KG_ENABLE_TEST_HOOKS=1 before its FIRST kg_scan* -- that
 *   one is read once, so a stray KG_TEST_* variable in a server's environment does nothing): KG_TEST_TINY_LISTS=1 starts the hit / candidate lists at one chunk, so that the
 *   resize-and-rerun path runs; KG_TEST_FAIL_ALLOC=n makes the n-th device allocation of the call fail with
 *   KG_ERR_NOMEM, so that the error paths can be checked for leaks (kg_table_live_device_bytes).
 */
#ifndef KMERGUTS_HIP_H
#define KMERGUTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_OK               0
#define KG_ERR_ARG         (-1)   /* bad argument                                                  */
#define KG_ERR_IO          (-2)   /* file could not be read                                        */
#define KG_ERR_FORMAT      (-3)   /* table image malformed (short header, entrySize != 24, ...)    */
#define KG_ERR_DEVICE      (-4)   /* HIP runtime error / no gfx950 device                          */
#define KG_ERR_NOMEM       (-5)
#define KG_ERR_UNSUPPORTED (-6)   /* parameters on which the reference itself throws (minHits < 2) */
#define KG_ERR_LIMIT       (-7)   /* one call exceeds 2^32-1 windows or 2^31-1 window blocks       */
#define KG_ERR_BUSY        (-8)   /* another kg_scan* is in flight on the same kg_table            */

/* KGJ:85-99 */
#define KG_K                8
#define KG_MAX_ENCODED      25600000000LL   /* 20^8; a table slot is empty iff whichKmer > this (KGJ:1000) */
#define KG_MAX_HITS_PER_SEQ 40000
#define KG_OI_BUFSZ         5
#define KG_TABLE_ENTRY_SIZE 24              /* KGJ:995-999: i64 whichKmer, i32 otuIndex, i32 avgFromEnd, i32 functionIndex, f32 functionWt */

/* The instance fields the hot path reads (KGJ:102-106), set by the CLI flags -a -O -m -M -g (KGJ:577-595). */
typedef struct kg_params {
    int32_t aa;                 /* -a : input is protein (1 container per sequence) instead of DNA (6)  */
    int32_t order_constraint;   /* -O */
    int32_t min_hits;           /* -m, reference default 5; must be >= 2                                */
    int32_t min_weighted_hits;  /* -M, reference default 0                                              */
    int32_t max_gap;            /* -g, reference default 200                                            */
    uint32_t flags;             /* KG_F_*                                                               */
} kg_params;

#define KG_F_COUNTERS        1u  /* also count windows_valid / slots_inspected, slower     */
#define KG_F_SKIP_AGGREGATE  2u  /* stop after the hit records (no CALL / OTU stage)                   */
#define KG_F_PROGRESS        4u  /* also record what the reference's table stream would have reported  */
                                 /* (kg_result_progress, kg_result_hit_slots): the "Processed: NN%"    */
                                 /* lines of KGJ:1016-1025 and where a short table file fails,          */
                                 /* KGJ:985-988 / 1036-1049.  Runs the walking kernels of KG_F_COUNTERS */
                                 /* (the stats' counters are filled too); tables of < 2^32 records      */

/* Event byte per hit record (kg_result_hit_events) and per container (kg_result_container_tail_events):
 * what gatherHits (KGJ:457-514) did at that record, so that a host can print the -d stream (HIT, after-hit,
 * after-call; KGJ:376-383, 406-409, 470-473, 498-501) by replaying the list contents without deciding anything.
 * Order of the steps at one record: [RESET_BEFORE] -> [ACCEPTED] -> [RESET_AFTER]. */
#define KG_EV_ACCEPTED      0x01u  /* the record was appended to the hits list (KGJ:496-497)                      */
#define KG_EV_RESET_BEFORE  0x02u  /* gap rule (KGJ:477-484): the list was processed or cleared before the record */
#define KG_EV_CALL_BEFORE   0x04u  /*   ... and that printed the container's next CALL                            */
#define KG_EV_KEEP2_BEFORE  0x08u  /*   ... and the list kept its last two members (KGJ:441-449), else it is empty */
#define KG_EV_RESET_AFTER   0x10u  /* pair rule (KGJ:503-508): processSetOfHits ran after the append step         */
#define KG_EV_CALL_AFTER    0x20u
#define KG_EV_KEEP2_AFTER   0x40u
#define KG_EV_TAIL_CALL     0x01u  /* container byte: the final flush (KGJ:511-513) printed a CALL                */

/* Binary records.  Text formatting (KGJ:398-404, 518-548) stays in host code. */
typedef struct kg_hit {          /* replaces class Hit (KGJ:1213-1219) + its HitContainer id (KGJ:1262-1266) */
    uint32_t container;          /* running container index: seq*6 + {+0,+1,+2,-0,-1,-2} (DNA) or seq (AA), KGJ:907-911 */
    int32_t  from0InProt;
    int32_t  oI;
    int32_t  avgOffFromEnd;
    int32_t  fI;
    float    functionWt;
} kg_hit;

typedef struct kg_call {         /* one "CALL" line, KGJ:398-404 */
    uint32_t container;
    int32_t  start;              /* hits.get(0).from0InProt                 */
    int32_t  end;                /* hits.get(lastHit).from0InProt + (K-1)   */
    int32_t  count;              /* fICount                                 */
    int32_t  fI;                 /* currentFI                               */
    float    weightedHits;       /* float32, summed in list order           */
} kg_call;

typedef struct kg_otu {          /* the per-sequence oICounts buffer as printed by KGJ:516-524 */
    int32_t n;
    int32_t count[KG_OI_BUFSZ];
    int32_t oI[KG_OI_BUFSZ];
} kg_otu;

typedef struct kg_stats {
    int64_t n_seqs, n_containers, n_blocks;   /* blocks = wavefront work items ("contig window blocks") */
    int64_t n_hits, n_calls;
    int64_t residues;            /* translated positions (DNA: sum over 6 frames) or characters (AA)     */
    int64_t windows;             /* 8-residue windows enumerated (KGJ:912 loop trips, all frames)        */
    int64_t windows_valid;       /* windows that encode (KGJ:913-915); valid only with KG_F_COUNTERS     */
    int64_t slots_inspected;     /* table entries inspected under KGJ:944-1034 semantics; KG_F_COUNTERS  */
    int64_t table_bytes;         /* numSigs * 24                                                          */
    float   ms_scan;             /* HIP-event time of the scan kernel (encode + probe + compaction)      */
    float   ms_order;            /* prefix sums + ordered placement of the hit records                   */
    float   ms_aggregate;        /* gatherHits / processSetOfHits kernels                                */
    float   ms_total;            /* first kernel start -> last kernel end on the library's stream        */
    int32_t scan_launches;       /* >1 when the hit staging buffer had to grow and the scan was re-run   */
    int32_t partitioned;         /* 1: the scan stage ran as scatter + tag + verify passes (kg_partition.hpp), */
                                 /* 0: as the single direct-probing kernel                                    */
    float   ms_part_scatter;     /* partitioned only: start of ms_scan until the last chunk is scattered        */
    float   ms_part_tag;         /* (unused: the tag passes overlap the scatter passes of later chunks)         */
    float   ms_part_verify;      /* partitioned only: what remains of the tag / verify passes after that        */
    int32_t fallback;            /* why a partitioned attempt was thrown away and the direct kernel ran instead:  */
                                 /* 0 none (partitioned ran, or direct was chosen up front), 1 more overflow      */
                                 /* groups than provisioned (heavily repeated k-mers), 2 the scatter pass's spin  */
                                 /* guard fired (protocol failure; never expected)                                */
    int32_t part_chunks;         /* partitioned only: chunks of whole sequences the batch was cut into            */
    int32_t part_buckets;        /* partitioned only: slot-range buckets (each 2^part_shift slots)                */
    int32_t part_shift;
    int32_t lookup_ran_off;      /* 1: some query walked to the end of the record stream undecided -- where the     */
                                 /* reference's table stream throws EOFException and its lookup ends with            */
                                 /* "Error: null" instead of "Kmers found: ..." (KGJ:799-802, 1031-1033, 1097-1126); */
                                 /* the records are the same either way (EOF == not found)                           */
    int32_t agg_pieces;          /* pieces beyond the first that long containers were cut into for gatherHits (cuts  */
                                 /* at gaps > maxGap, where the reference's list restarts anyway: KGJ:477-484)       */
    int32_t part_levels;         /* partitioned only: what the tag pass probed in the L2 -- 1 = the tags (bucket_tag_kernel: scans  */
                                 /* with KG_F_COUNTERS, KG_BIDX=0), 4 = the table's byte home index (bucket_index_kernel, the       */
                                 /* default).  (2 and 3 were round 3's second partition level, removed in round 4.)                 */
} kg_stats;

/* KG_F_PROGRESS: the slots the reference's merge-join (KGJ:959-1029) visits = the slots some query's walk reads, summed up
 * the way its progress lines and its failure modes need them.  The join runs in slot order and prints
 *     "Processed: <10 f>%, time=<ms> ms., found-so-far=<k-mers found at slots <= s>"
 * at every visited slot s whose tenth f = (int)(10.0 * ((double)(s + 1) / (double)numSigs)) differs from the last one printed
 * (KGJ:1017-1024), i.e. once per tenth, at the first slot visited in it. */
typedef struct kg_progress {
    int64_t first_visited[11];   /* [f]: the first slot visited in tenth f, -1 when none (f = 10: the slot numSigs - 1 and, for   */
                                 /* a table file longer than numSigs records, everything behind it)                               */
    int64_t last_visited;        /* the last slot visited, -1 when none                                                           */
    int64_t first_beyond;        /* the smallest home slot of a query k-mer at or behind the END of the record stream (a table    */
                                 /* file shorter than numSigs records), -1 when none: the join has to skip to it and fails --     */
                                 /* "Error skipping <24 x (first_beyond - last_visited - 1)> bytes" on a .gz stream when the slot */
                                 /* lies BEHIND the end (first_beyond > stream_slots; KGJ:1036-1049), EOFException ("Error: null") */
                                 /* at the read that follows the skip on a plain file, or when the slot is the end itself         */
    int64_t walk_ran_off;        /* 1: a walk reached the end of the stream undecided: EOFException before any such skip          */
    int64_t stream_slots;        /* records in the table stream (numSigs for a complete file)                                     */
    int64_t found_upto[11];      /* [f]: distinct k-mers found at slots <= first_visited[f] = the line's found-so-far (0 when the */
                                 /* tenth was not visited)                                                                        */
    int64_t kmers_found;         /* distinct k-mers found by this scan = the reference's kmersFound (KGJ:1004-1006, 1031-1033)     */
} kg_progress;

typedef struct kg_table  kg_table;
typedef struct kg_result kg_result;

/* ---- signature table: replaces readKmerTableHeader (KGJ:924-942) + the table stream of lookup (KGJ:944-1034) ---- */

/* Read <path> = kmer.table.mem_map or kmer.table.mem_map.gz (KGJ:749-753; told apart by the gzip magic), validate the
 * header (3 x int64 LE: numSigs, entrySize, version) and make the table resident on HIP device <device>.  The file is
 * streamed through pinned buffers (several reader threads for a plain file, one zlib stream for .gz): no host copy. */
int kg_table_open(const char *path, int device, kg_table **out);
/* Same from a file image in host memory (the host gunzips kmer.table.mem_map.gz, KGJ:750-753). */
int kg_table_from_memory(const void *image, size_t nbytes, int device, kg_table **out);
/* Adopt num_sigs 24-byte entries that already sit in device memory (not copied, not freed; must stay
 * unchanged while the table lives).  Synchronises the device once before reading them. */
int kg_table_from_device(const void *d_entries, int64_t num_sigs, int device, kg_table **out);
/* header fields (KmerMemoryInfo, KGJ:1194-1198) and the number of occupied slots */
int kg_table_info(const kg_table *t, int64_t *num_sigs, int64_t *entry_size, int64_t *version, int64_t *occupied);
/* Bytes of device scratch / result blocks the table's block cache has handed out and not got back: the blocks of the
 * results still open, 0 when there are none (also after a failed kg_scan*: nothing may be left behind). */
int64_t kg_table_live_device_bytes(kg_table *t);
void kg_table_close(kg_table *t);

/* ---- the hot path: replaces prepareQuery/addKmers (KGJ:1051-1074, 900-922), the query sort
 *      (KGJ:1076-1095), lookup (KGJ:944-1034) and gatherHits/processSetOfHits (KGJ:385-514) for a
 *      batch of sequences.  seq = the raw concatenated sequence characters exactly as readFasta
 *      hands them to prepareQuery (KGJ:780-783); offsets[n_seqs+1] in host memory. ---- */
int kg_scan(kg_table *t, const kg_params *p, const uint8_t *seq, const int64_t *offsets,
            int64_t n_seqs, kg_result **out);
/* same with the sequence bytes already in device memory (offsets stay on the host).  The bytes must be
 * complete before the call: the library works on its own non-blocking stream. */
int kg_scan_device(kg_table *t, const kg_params *p, const uint8_t *d_seq, const int64_t *offsets,
                   int64_t n_seqs, kg_result **out);

/* ---- the aggregation alone: replaces the public gatherHits / processSetOfHits / processAASeq (KGJ:385-514, 526-536) for
 *      callers that hold hit records of their own.  hits[] ordered by (container, from0InProt), container_hit_start[n_seqs *
 *      (aa ? 1 : 6) + 1]; otu_init = the oICounts buffer every sequence starts with (n_seqs records) or NULL for empty
 *      buffers (KGJ:528, 540).  The result has the hit, CALL, OTU and event arrays (no table, no scan statistics). ---- */
int kg_aggregate_hits(int device, const kg_params *p, const kg_hit *hits, const int64_t *container_hit_start, int64_t n_seqs,
                      const kg_otu *otu_init, kg_result **out);

/* One step of that state machine, the public processSetOfHits (KGJ:385-455): votes for current_fi among hits[0..n_hits),
 * *called / *call = whether a CALL was made and its record, *otu = the caller's oICounts buffer (updated in place),
 * *new_current_fi = the method's return value, *keeps_last_two = the list keeps hits[n-2], hits[n-1] (else it is emptied). */
int kg_process_set_of_hits(int device, const kg_params *p, const kg_hit *hits, int32_t n_hits, int32_t current_fi, kg_otu *otu,
                           kg_call *call, int32_t *called, int32_t *new_current_fi, int32_t *keeps_last_two);

int kg_result_stats(const kg_result *r, kg_stats *out);
/* Host views, copied from the device on first use; NULL on failure (see kg_last_error). */
const kg_hit  *kg_result_hits(kg_result *r);                 /* n_hits, ordered by (container, from0InProt)   */
const int64_t *kg_result_container_hit_start(kg_result *r);  /* n_containers + 1                              */
const kg_call *kg_result_calls(kg_result *r);                /* n_calls, in the reference's emission order     */
const int64_t *kg_result_container_call_start(kg_result *r); /* n_containers + 1                              */
const kg_otu  *kg_result_otu(kg_result *r);                  /* n_seqs                                        */
const uint8_t *kg_result_hit_events(kg_result *r);           /* n_hits bytes of KG_EV_*                        */
const uint8_t *kg_result_container_tail_events(kg_result *r);/* n_containers bytes of KG_EV_TAIL_*             */
/* hits[first .. first + count) straight into caller-owned memory (e.g. a JNA Memory / numpy array); pageable
 * destinations are fed through the library's cached pinned blocks, the copy overlapped with the transfer. */
int kg_result_copy_hits(kg_result *r, int64_t first, int64_t count, kg_hit *dst);
/* KG_F_PROGRESS scans only (NULL / KG_ERR_ARG otherwise): the table slot every hit record was found at (n_hits entries,
 * parallel to kg_result_hits: distinct slots = distinct k-mers found, the reference's kmersFound, KGJ:1004-1006), and the
 * summary above.  Several scans of one run (batches): the slots combine by minimum / maximum; the two counts are per scan (a
 * k-mer found in two batches counts in both), a front end that needs them over several scans counts the distinct hit slots. */
const uint32_t *kg_result_hit_slots(kg_result *r);
int kg_result_progress(const kg_result *r, kg_progress *out);
/* Device views (valid until kg_result_free) for callers that keep working in HBM. */
const void    *kg_result_device_hits(const kg_result *r);
const void    *kg_result_device_calls(const kg_result *r);
const void    *kg_result_device_otu(const kg_result *r);
const void    *kg_result_device_container_hit_start(const kg_result *r);   /* int64[n_containers + 1] */
const void    *kg_result_device_container_call_start(const kg_result *r);  /* int64[n_containers + 1], NULL with KG_F_SKIP_AGGREGATE */
void kg_result_free(kg_result *r);

/* ---- multi-GPU exchange helper (no counterpart in the reference, which is one process; used by the host layer that
 *      gathers per-rank hit buffers, kmergutsjava_amd/distributed.py).  The n_hits records at d_src are ordered by a shard's
 *      LOCAL containers; the records of local sequence k are d_src[d_seq_first[k] .. d_seq_first[k + 1]) (d_seq_first has
 *      n_seqs + 1 entries) and go to d_dst[d_dst_first[k] ..) with d_container_shift[k] added to their container field.
 *      All pointers are device memory; the copy is enqueued on `stream` (a hipStream_t, NULL = the null stream) and not
 *      waited for. ---- */
int kg_restore_hits_device(int device, const kg_hit *d_src, int64_t n_hits, const int64_t *d_seq_first, int64_t n_seqs,
                           const int64_t *d_dst_first, const int32_t *d_container_shift, kg_hit *d_dst, void *stream);

const char *kg_last_error(void);
/* "libkmerguts_hip <version> gfx950" */
const char *kg_version(void);

#ifdef __cplusplus
}
#endif
#endif
