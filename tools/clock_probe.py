#!/usr/bin/env python3
"""Shader clock / power while the 1 Gbp scan runs back to back (is the plateau a power or clock limit?).
Samples `rocm-smi` in a child process every ~0.2 s during ~5 s of scans.  Tuning aid."""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth

num_sigs, total_bp = 1400303159, 1000000000
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev)
torch.cuda.synchronize()
samples, stop = [], False

def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), json.loads(out)))
        except Exception as e:
            samples.append((time.perf_counter(), {"error": str(e)}))
        time.sleep(0.15)

idle = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); times = []
for i in range(200):
    with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
        times.append(r.stats["ms_total"])
wall = time.perf_counter() - t0
stop = True; th.join()
print(json.dumps({"idle": json.loads(idle) if idle.strip().startswith("{") else idle[:300]}))
for t, s in samples:
    print(json.dumps({"t": round(t - t0, 2), "smi": s}))
print(json.dumps({"scans": len(times), "wall_s": wall, "ms_total_first": times[0], "ms_total_median": sorted(times)[len(times) // 2],
                  "ms_total_last": times[-1]}))
