import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
dev = torch.device("cuda", 0)
num_sigs = 1400303159
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
lens = synth.contig_mix_lengths(1_000_000_000, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev)
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
P = hotpath.Params()
rows = []
for rep in range(8):
    t0 = time.perf_counter()
    r = tab.scan(None, off, P, device_ptr=seq.data_ptr())
    t1 = time.perf_counter()
    st = r.stats
    t2 = time.perf_counter()
    r.calls(copy=False)
    t3 = time.perf_counter()
    r.otu(copy=False)
    t4 = time.perf_counter()
    r.close()
    t5 = time.perf_counter()
    rows.append([st["ms_total"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3])
for r_ in rows[2:]:
    print(json.dumps(dict(zip(["ms_total", "scan_call", "stats", "calls", "otu", "close", "step"], [round(x, 3) for x in r_]))))
