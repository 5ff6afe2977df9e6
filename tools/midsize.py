#!/usr/bin/env python3
"""Mid-size batches against the full table: BASELINE config 2 (1000 x 100 kbp) and the shards a strong-scaling rank of
bench.py --gpus N gets (1 Gbp contig mix / N).  One JSON line per case: wall time of the bench step (scan + CALL / OTU
records to the host) and the library's stage times.  SW_CASES picks cases, e.g. "c2,8" (c2, 16, 8, 4, 2, 1)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth, distributed as kd
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159"))
reps = int(os.environ.get("SW_REPS", "6"))
cases = os.environ.get("SW_CASES", "c2,8,4,2").split(",")
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
all_lens = synth.contig_mix_lengths(1_000_000_000, 301); all_off = synth.offsets_of(all_lens)
for case in cases:
    if case == "c2":
        seq, off = synth.dna_uniform_config(1000, 100_000, 201, dev)
    else:
        mine = kd.shard_sequences(all_lens, int(case))[0]
        lens = all_lens[mine]; off = synth.offsets_of(lens)
        seq = synth.random_dna_at(all_off[mine], lens, 302, dev)
    torch.cuda.synchronize()
    rows = []
    for rep in range(reps):
        t0 = time.perf_counter()
        with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
            r.calls(); r.otu(); st = r.stats
        rows.append(dict(st, wall_ms=(time.perf_counter() - t0) * 1e3))
    best = min(rows[1:], key=lambda x: x["wall_ms"])
    print(json.dumps({"case": case, "bp": int(off[-1]), "contigs": len(off) - 1, "wall_ms": best["wall_ms"],
                      "residues_per_s": best["residues"] / best["wall_ms"] * 1e3,
                      "ms_scan": best["ms_scan"], "ms_order": best["ms_order"], "ms_aggregate": best["ms_aggregate"],
                      "ms_total": best["ms_total"], "partitioned": best["partitioned"], "chunks": best["part_chunks"],
                      "n_hits": best["n_hits"], "all_wall_ms": [round(x["wall_ms"], 3) for x in rows]}), flush=True)
