package kmergutsjava;

import com.sun.jna.Library;
import com.sun.jna.Native;
import com.sun.jna.Pointer;
import com.sun.jna.Structure;
import com.sun.jna.ptr.IntByReference;
import com.sun.jna.ptr.PointerByReference;

/**
 * JNA binding of libkmerguts_hip.so (C ABI: include/kmerguts_hip.h), the MI355X implementation of the
 * kmer_guts hot path.  jna-3.4.0.jar is already on the reference's classpath (build.xml:27), so a
 * maintainer adds this one file and the call site shown in INTEGRATION.md.
 *
 * Written against the JNA level the reference ships, 3.4.0: a Structure's field order is given with
 * setFieldOrder(String[]) in its constructor (the abstract getFieldOrder() only exists from JNA 3.5.0 on),
 * Native.loadLibrary(String, Class) loads the library.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no JDK there); it is kept mechanical on purpose:
 * one Java method per exported C function, structures field for field -- tests/test_java_binding.py parses this
 * file and checks method names, arities and structure fields against include/kmerguts_hip.h.
 */
public interface KmerGutsHip extends Library {
    KmerGutsHip LIB = (KmerGutsHip) Native.loadLibrary("kmerguts_hip", KmerGutsHip.class);

    int KG_OK = 0;
    int KG_ERR_ARG = -1, KG_ERR_IO = -2, KG_ERR_FORMAT = -3, KG_ERR_DEVICE = -4, KG_ERR_NOMEM = -5, KG_ERR_UNSUPPORTED = -6,
        KG_ERR_LIMIT = -7, KG_ERR_BUSY = -8;
    int KG_F_COUNTERS = 1;
    int KG_F_SKIP_AGGREGATE = 2;
    int KG_F_PROGRESS = 4;

    /** struct kg_params: the instance fields the hot path reads (KmerGutsJava.java:102-106). */
    class KgParams extends Structure {
        public int aa, order_constraint, min_hits, min_weighted_hits, max_gap, flags;
        public KgParams() {
            setFieldOrder(new String[] {"aa", "order_constraint", "min_hits", "min_weighted_hits", "max_gap", "flags"});
        }
    }

    /** struct kg_stats */
    class KgStats extends Structure {
        public long n_seqs, n_containers, n_blocks, n_hits, n_calls, residues, windows, windows_valid,
                slots_inspected, table_bytes;
        public float ms_scan, ms_order, ms_aggregate, ms_total;
        public int scan_launches, partitioned;
        public float ms_part_scatter, ms_part_tag, ms_part_verify;
        public int fallback, part_chunks, part_buckets, part_shift, lookup_ran_off, agg_pieces, part_levels;
        public KgStats() {
            setFieldOrder(new String[] {"n_seqs", "n_containers", "n_blocks", "n_hits", "n_calls", "residues", "windows",
                    "windows_valid", "slots_inspected", "table_bytes", "ms_scan", "ms_order", "ms_aggregate",
                    "ms_total", "scan_launches", "partitioned", "ms_part_scatter", "ms_part_tag", "ms_part_verify", "fallback",
                    "part_chunks", "part_buckets", "part_shift", "lookup_ran_off", "agg_pieces", "part_levels"});
        }
    }

    /** struct kg_progress (KG_F_PROGRESS scans): what lookup's table stream reports and where it fails (KmerGutsJava.java:1016-1049). */
    class KgProgress extends Structure {
        public long[] first_visited = new long[11];
        public long last_visited, first_beyond, walk_ran_off, stream_slots;
        public long[] found_upto = new long[11];
        public long kmers_found;
        public KgProgress() {
            setFieldOrder(new String[] {"first_visited", "last_visited", "first_beyond", "walk_ran_off", "stream_slots", "found_upto",
                    "kmers_found"});
        }
    }

    // replaces readKmerTableHeader + the table stream of lookup (KmerGutsJava.java:924-942, 944-1034)
    int kg_table_open(String path, int device, PointerByReference out);
    int kg_table_from_memory(Pointer image, long nbytes, int device, PointerByReference out);
    int kg_table_from_device(Pointer dEntries, long numSigs, int device, PointerByReference out);
    int kg_table_info(Pointer table, long[] numSigs, long[] entrySize, long[] version, long[] occupied);
    long kg_table_live_device_bytes(Pointer table);
    void kg_table_close(Pointer table);

    // replaces prepareQuery/addKmers, the query sort, lookup and gatherHits/processSetOfHits
    // (KmerGutsJava.java:1051-1074, 900-922, 1076-1095, 944-1034, 385-514) for a batch of sequences
    int kg_scan(Pointer table, KgParams params, byte[] seq, long[] offsets, long nSeqs, PointerByReference out);
    int kg_scan_device(Pointer table, KgParams params, Pointer dSeq, long[] offsets, long nSeqs, PointerByReference out);

    int kg_aggregate_hits(int device, KgParams params, Pointer hits, Pointer containerHitStart, long nSeqs, Pointer otuInit,
                          PointerByReference out);
    int kg_process_set_of_hits(int device, KgParams params, Pointer hits, int nHits, int currentFI, Pointer otu, Pointer call,
                               IntByReference called, IntByReference newCurrentFI, IntByReference keepsLastTwo);
    int kg_result_stats(Pointer result, KgStats out);
    Pointer kg_result_hits(Pointer result);                  // kg_hit[n_hits]   24 B each
    Pointer kg_result_container_hit_start(Pointer result);   // int64[n_containers + 1]
    Pointer kg_result_calls(Pointer result);                 // kg_call[n_calls] 24 B each
    Pointer kg_result_container_call_start(Pointer result);  // int64[n_containers + 1]
    Pointer kg_result_otu(Pointer result);                   // kg_otu[n_seqs]   44 B each
    Pointer kg_result_hit_events(Pointer result);            // byte[n_hits]        KG_EV_* (for the -d stream)
    Pointer kg_result_container_tail_events(Pointer result); // byte[n_containers]  KG_EV_TAIL_CALL
    Pointer kg_result_device_hits(Pointer result);
    Pointer kg_result_device_calls(Pointer result);
    int kg_result_copy_hits(Pointer result, long first, long count, Pointer dst);
    Pointer kg_result_hit_slots(Pointer result);             // uint32[n_hits]      KG_F_PROGRESS scans
    int kg_result_progress(Pointer result, KgProgress out);
    Pointer kg_result_device_otu(Pointer result);
    Pointer kg_result_device_container_hit_start(Pointer result);
    Pointer kg_result_device_container_call_start(Pointer result);
    void kg_result_free(Pointer result);
    int kg_restore_hits_device(int device, Pointer dSrc, long nHits, Pointer dSeqFirst, long nSeqs, Pointer dDstFirst,
                               Pointer dContainerShift, Pointer dDst, Pointer stream);

    String kg_last_error();
    String kg_version();
}
