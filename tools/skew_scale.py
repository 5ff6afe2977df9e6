#!/usr/bin/env python3
"""Low-complexity input at scale: 400 Mbp contig mix with homopolymer / dinucleotide runs spliced in (5 % of the bases),
both strategies: same records?  how much slower?  Sanity / tuning aid."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth

num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159"))
total_bp = int(os.environ.get("SW_TOTAL_BP", "400000000"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev)
# splice runs: every 20th contig longer than 50 kbp gets a 50 kbp run of A, the next one of "AT" repeats
runs = 0
for k in range(len(lens)):
    if lens[k] >= 100000 and k % 4 == 0:
        a = int(off[k]) + 1000
        if runs % 2 == 0:
            seq[a:a + 50000] = ord("A")
        else:
            seq[a:a + 50000] = torch.tensor([ord("A"), ord("T")], dtype=torch.uint8, device=dev).repeat(25000)
        runs += 1
torch.cuda.synchronize()
res = {}
for mode in (0, 1):
    os.environ["KG_PARTITION"] = str(mode)
    best = None
    for rep in range(3):
        with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
            st = r.stats
            if rep == 2:
                h = r.hits(); sig = (st["n_hits"], st["n_calls"], int(h["fI"].astype(np.int64).sum()), int(h["from0InProt"].astype(np.int64).sum()))
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    res[mode] = sig
    print(json.dumps({"KG_PARTITION": mode, "partitioned": best["partitioned"], "ms_scan": best["ms_scan"], "ms_total": best["ms_total"],
                      "runs_of_50kbp": runs, "n_hits": best["n_hits"], "signature": sig}), flush=True)
assert res[0] == res[1], res
print(json.dumps({"identical_hit_signature": True}))
