#!/usr/bin/env python3
"""PCIe-inclusive rate: the same 1 Gbp workload handed over as HOST buffers through kg_scan (pageable H2D of the
sequence bytes, CALL/OTU records back).  DESIGN.md section 8.  Not the bench `value`."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159")); total_bp = int(os.environ.get("SW_TOTAL_BP", "1000000000"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev); torch.cuda.synchronize()
host = seq.cpu().numpy()
out = {}
for name, fn in (("device_input", lambda: tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr())),
                 ("host_input", lambda: tab.scan(host, off, hotpath.Params()))):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        with fn() as r:
            r.calls(); r.otu(); st = r.stats
        ts.append(time.perf_counter() - t0)
    out[name] = {"ms_per_step": min(ts[1:]) * 1e3, "residues_per_s": st["residues"] / min(ts[1:]), "ms_scan": st["ms_scan"]}
t0 = time.perf_counter()
with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
    h = r.hits()
out["hits_to_host"] = {"n_hits": len(h), "ms_scan_plus_copy": (time.perf_counter() - t0) * 1e3}
print(json.dumps(out))
