#!/usr/bin/env python3
"""Experiment: TWO scans in flight on one GPU (two table handles over the same records in HBM, one host thread each) against
one after the other -- what a caller with a stream of batches could gain.  SW_CASES as tools/midsize.py ("c2", "8", ..., "1" = 1 Gbp).
One JSON line per case: ms per scan back to back, and with two in flight."""
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth, distributed as kd
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159"))
n_scans = int(os.environ.get("SW_SCANS", "12"))
cases = os.environ.get("SW_CASES", "c2,8").split(",")
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tabs = [hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec) for _ in range(2)]
all_lens = synth.contig_mix_lengths(1_000_000_000, 301); all_off = synth.offsets_of(all_lens)


def run(tab, seq, off, n, out, delay=0.0):
    if delay:
        time.sleep(delay)                              # SW_STAGGER_MS: the second thread starts in the middle of the first one's scan
    for _ in range(n):
        with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
            r.calls(); r.otu(); out.append(r.stats["n_hits"])


for case in cases:
    if case == "c2":
        seq, off = synth.dna_uniform_config(1000, 100_000, 201, dev)
    else:
        mine = kd.shard_sequences(all_lens, int(case))[0]
        lens = all_lens[mine]; off = synth.offsets_of(lens)
        seq = synth.random_dna_at(all_off[mine], lens, 302, dev)
    torch.cuda.synchronize()
    for t in tabs:
        run(t, seq, off, 2, [])                       # warm both handles
    torch.cuda.synchronize()
    t0 = time.perf_counter(); h1 = []
    run(tabs[0], seq, off, n_scans, h1)
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / n_scans * 1e3
    t0 = time.perf_counter(); h2 = [[], []]
    stagger = float(os.environ.get("SW_STAGGER_MS", "0")) * 1e-3
    th = [threading.Thread(target=run, args=(tabs[k], seq, off, n_scans // 2, h2[k], stagger * k)) for k in range(2)]
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / (2 * (n_scans // 2)) * 1e3
    assert set(h1) == set(h2[0]) == set(h2[1])
    print(json.dumps({"case": case, "bp": int(off[-1]), "ms_per_scan_one_at_a_time": serial, "ms_per_scan_two_in_flight": two}), flush=True)
