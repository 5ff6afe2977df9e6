#!/usr/bin/env python3
"""Scale diagnostics: counters and hit totals at several sizes / grids (debug aid)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
dev = torch.device("cuda", 0)
for num_sigs, total_bp in ((10_000_019, 20_000_000), (100_000_007, 100_000_000), (1_400_303_159, 200_000_000)):
    rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
    torch.cuda.synchronize()
    tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
    lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 302, dev)
    torch.cuda.synchronize()
    for grid in (16384, 2048, 256):
        for rpg in (3, 6):
            os.environ["KG_SCAN_GRID"] = str(grid); os.environ["KG_SCAN_RPG"] = str(rpg)
            with tab.scan(None, off, hotpath.Params(counters=True), device_ptr=seq.data_ptr()) as r:
                st = r.stats
            print(json.dumps(dict(num_sigs=num_sigs, bp=total_bp, grid=grid, rpg=rpg, placed=placed, info=tab.info(),
                                  n_blocks=st["n_blocks"], n_hits=st["n_hits"], valid=st["windows_valid"],
                                  slots=st["slots_inspected"], ms_scan=st["ms_scan"], launches=st["scan_launches"])), flush=True)
    tab.close(); del rec, seq
