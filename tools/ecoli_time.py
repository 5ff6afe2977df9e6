#!/usr/bin/env python3
"""The reference's E. coli genome (one 4.6 Mbp contig, tests/golden) against the synthetic table of the parity test
(400 k of the proteome's own 8-mers + 400 k random signatures) and against a denser one: stage times of the scan
(tuning aid: one long contig = six long containers for gatherHits)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmergutsjava_amd import hotpath, synth
from test_gpu_parity import _ecoli, _img
ids_p, prot, off_p = _ecoli("Ecoli_K12_W3110.faa.gz")
ids_g, dna, off_g = _ecoli("Ecoli_K12_W3110.fna.gz")
codes = synth.aa_codes(torch.frombuffer(bytearray(prot), dtype=torch.uint8))
vals = synth.encode_windows_aa(codes); vals = vals[vals >= 0]
for n_own in [int(x) for x in os.environ.get("SW_OWN", "400000,3000000").split(",")]:
    own = vals[synth._uniform(71, 0, n_own, int(vals.numel()), "cpu")]
    keys = torch.unique(torch.cat([own, synth.random_keys(400000, 72)]))
    fn = (synth._lsr(synth.splitmix64(73, keys // 20 ** 5), 3) % 500).to(torch.int32)
    otu, avg, _, wt = synth.payload_of(keys, 74, n_otu=12)
    rec, placed = synth.build_table(keys, (otu, avg, fn, wt), 8_000_009)
    with hotpath.SignatureTable.from_bytes(_img(rec)) as tab:
        for rep in range(4):
            t0 = time.perf_counter()
            with tab.scan(dna, off_g, hotpath.Params(min_hits=3)) as r:
                r.calls(); st = r.stats
            wall = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"own_8mers": n_own, "wall_ms": wall, "ms_scan": st["ms_scan"], "ms_order": st["ms_order"],
                          "ms_aggregate": st["ms_aggregate"], "n_hits": st["n_hits"], "n_calls": st["n_calls"],
                          "longest_container": int(np.diff(r.container_hit_start()).max()) if False else None}), flush=True)
# the proteome (13 645 proteins, -a)
own = vals[synth._uniform(71, 0, 400_000, int(vals.numel()), "cpu")]
keys = torch.unique(torch.cat([own, synth.random_keys(400000, 72)]))
fn = (synth._lsr(synth.splitmix64(73, keys // 20 ** 5), 3) % 500).to(torch.int32)
otu, avg, _, wt = synth.payload_of(keys, 74, n_otu=12)
rec, placed = synth.build_table(keys, (otu, avg, fn, wt), 8_000_009)
with hotpath.SignatureTable.from_bytes(_img(rec)) as tab:
    for rep in range(4):
        t0 = time.perf_counter()
        with tab.scan(prot, off_p, hotpath.Params(aa=True)) as r:
            r.calls(); st = r.stats
        wall = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"proteome": True, "wall_ms": wall, "ms_scan": st["ms_scan"], "ms_order": st["ms_order"], "ms_aggregate": st["ms_aggregate"],
                      "n_hits": st["n_hits"], "n_calls": st["n_calls"]}), flush=True)
