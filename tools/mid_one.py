#!/usr/bin/env python3
"""One mid-size batch (the 125 Mbp shard of one of 8 ranks, or SW_CASE=c2: config 2) scanned a few times, for kernel traces."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth, distributed as kd
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159")); reps = int(os.environ.get("SW_REPS", "4")); case = os.environ.get("SW_CASE", "8")
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
if case == "c2":
    seq, off = synth.dna_uniform_config(1000, 100_000, 201, dev)
else:
    all_lens = synth.contig_mix_lengths(1_000_000_000, 301); all_off = synth.offsets_of(all_lens)
    mine = kd.shard_sequences(all_lens, int(case))[0]
    lens = all_lens[mine]; off = synth.offsets_of(lens)
    seq = synth.random_dna_at(all_off[mine], lens, 302, dev)
torch.cuda.synchronize()
for rep in range(reps):
    t0 = time.perf_counter()
    with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
        r.calls(); r.otu(); st = r.stats
    print(json.dumps(dict(st, wall_ms=(time.perf_counter() - t0) * 1e3)), flush=True)
