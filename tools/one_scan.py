#!/usr/bin/env python3
"""Full-size scan, a few repetitions, for rocprofv3 kernel traces (debug/tuning aid)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159")); total_bp = int(os.environ.get("SW_TOTAL_BP", "1000000000"))
reps = int(os.environ.get("SW_REPS", "3"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev); torch.cuda.synchronize()
for rep in range(reps):
    t0 = time.perf_counter()
    with tab.scan(None, off, hotpath.Params(counters=bool(int(os.environ.get("SW_COUNTERS", "0")))), device_ptr=seq.data_ptr()) as r:
        st = r.stats
    print(json.dumps(dict(st, wall_ms=(time.perf_counter() - t0) * 1e3)), flush=True)
