#!/usr/bin/env python3
"""Sweep the scan kernel's tunables (rows per group, staging chunk, persistent grid) at full size.
Writes one JSON line per variant to stdout.  Tuning aid, not part of the product path."""
import itertools, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth

num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159"))
total_bp = int(os.environ.get("SW_TOTAL_BP", "1000000000"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev)
torch.cuda.synchronize()
variants = [dict(KG_PARTITION=0)]
if os.environ.get("SW_VARIANTS"):            # e.g. SW_VARIANTS='[{"KG_PROBE_GRID": 2048}, ...]' (KG_PARTITION=1 implied)
    variants = [dict(dict(KG_PARTITION=1), **v) for v in json.loads(os.environ["SW_VARIANTS"])]
else:
    for qg in (1024, 1792, 2048, 3584):
        variants.append(dict(KG_PARTITION=1, KG_PART_SHIFT=21, KG_PROBE_GRID=qg, KG_VERIFY_GRID=2048))
    for shift in (20, 22):
        variants.append(dict(KG_PARTITION=1, KG_PART_SHIFT=shift, KG_PROBE_GRID=1792, KG_VERIFY_GRID=2048))
    variants.append(dict(KG_PARTITION=1, KG_PART_SHIFT=21, KG_PROBE_GRID=1792, KG_VERIFY_GRID=4096))
ref = None
for v in variants:
    for k, x in v.items():
        os.environ[k] = str(x)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
            st = r.stats
            wall = (time.perf_counter() - t0) * 1e3
            if rep == 2:
                sig = (st["n_hits"], st["n_calls"])
        if best is None or st["ms_scan"] < best["ms_scan"]:
            best = dict(st, wall_ms=wall)
    if ref is None:
        ref = sig
    assert sig == ref, (sig, ref)
    print(json.dumps(dict(v, ms_scan=best["ms_scan"], ms_order=best["ms_order"], ms_aggregate=best["ms_aggregate"],
                          ms_total=best["ms_total"], wall_ms=best["wall_ms"], launches=best["scan_launches"],
                          n_hits=best["n_hits"], part_scatter=best.get("ms_part_scatter"), part_tail=best.get("ms_part_verify"),
                          lib=os.environ.get("KG_LIB_PATH", ""))), flush=True)
