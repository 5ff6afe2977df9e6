#!/usr/bin/env python3
"""BASELINE config 5 alone (100 Mbp assembled from signature k-mers, <= 32 functions / <= 8 OTUs): a few timed steps, for
kernel traces of the aggregation stage.  SW_DNA=0 for the protein variant."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
dna = bool(int(os.environ.get("SW_DNA", "1")))
dev = torch.device("cuda", 0)
n_contigs, kpc = (1000, 4167) if dna else (10000, 38)
seq, off, rec = synth.high_density_device(n_contigs, kpc, 20_000_003, 8_000_000, 501, dna, dev)
torch.cuda.synchronize()
with hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), 20_000_003, 0, keepalive=rec) as tab:
    rows = []
    for rep in range(int(os.environ.get("SW_REPS", "6"))):
        t0 = time.perf_counter()
        with tab.scan(None, off, hotpath.Params(aa=not dna), device_ptr=seq.data_ptr()) as r:
            r.calls(copy=False); r.otu(copy=False); st = r.stats
        rows.append(dict(st, wall_ms=(time.perf_counter() - t0) * 1e3))
    best = min(rows[1:], key=lambda x: x["wall_ms"])
    print(json.dumps({k: best[k] for k in ("wall_ms", "ms_scan", "ms_order", "ms_aggregate", "ms_total", "n_hits", "n_calls", "residues", "partitioned")}))
