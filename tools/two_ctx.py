#!/usr/bin/env python3
"""Experiment: two table handles over ONE device copy of the signature records, two host threads scanning batches
back to back -- does the head of one scan fill the tail of the other?  (tuning aid; SW_THREADS=1 is the control)"""
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159")); total_bp = int(os.environ.get("SW_TOTAL_BP", "1000000000"))
reps = int(os.environ.get("SW_REPS", "6")); n_thr = int(os.environ.get("SW_THREADS", "2"))
stagger_ms = float(os.environ.get("SW_STAGGER_MS", "0"))
period_ms = float(os.environ.get("SW_PERIOD_MS", "0"))      # > 0: scan i (thread i % n) starts at t0 + i * period
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tabs = [hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec) for _ in range(n_thr)]
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev); torch.cuda.synchronize()
params = hotpath.Params()
sums = [None] * n_thr

def one(tab):
    r = tab.scan(None, off, params, device_ptr=seq.data_ptr())
    st = r.stats
    r.calls(copy=False); r.otu(copy=False)
    r.close()
    return st

for t in tabs:                    # warm-up: buffers sized
    one(t); one(t)
torch.cuda.synchronize()
go = threading.Barrier(n_thr + 1)

t_start = [0.0]

def worker(k):
    go.wait()
    if k and stagger_ms: time.sleep(stagger_ms * 1e-3)
    ms = []
    for j in range(reps):
        if period_ms:
            due = t_start[0] + (j * n_thr + k) * period_ms * 1e-3
            while time.perf_counter() < due: pass
        ta = time.perf_counter()
        st = one(tabs[k])
        ms.append((time.perf_counter() - ta) * 1e3 if period_ms else st["ms_scan"])
    sums[k] = (st["n_hits"], st["n_calls"], ms)

th = [threading.Thread(target=worker, args=(k,)) for k in range(n_thr)]
for x in th: x.start()
t_start[0] = time.perf_counter() + 0.002
go.wait()
t0 = time.perf_counter()
for x in th: x.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"threads": n_thr, "reps_each": reps, "ms_per_scan_aggregate": el * 1e3 / (reps * n_thr), "stagger_ms": stagger_ms, "period_ms": period_ms,
                  "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "order_streams": os.environ.get("KG_ORDER_STREAMS"),
                  "hits": [s[0] for s in sums], "calls": [s[1] for s in sums],
                  "ms_scan_seen": [[round(x, 2) for x in s[2]] for s in sums]}), flush=True)
