#!/usr/bin/env python3
"""What rank 0 pays per step to put the gathered hit records of N ranks into global order (distributed.restore_hits), at
BASELINE config 4's size: the 1 Gbp contig list scanned once, its hits cut into the N shards' buffers as the ranks would
send them, then restored and compared with the unsharded records.  One JSON line per N."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import distributed as kd, hotpath, synth
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159")); total_bp = int(os.environ.get("SW_TOTAL_BP", "1000000000"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
lens = synth.contig_mix_lengths(total_bp, 301); off = synth.offsets_of(lens)
seq = synth.random_dna(int(off[-1]), 302, dev); torch.cuda.synchronize()
with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
    hits = r.device_view("hits").clone().view(torch.int32).view(-1, 6)
    chs = r.device_view("container_hit_start").clone()
per = 6
for world in (2, 4, 8):
    shards = kd.shard_sequences(lens, world)
    hb, cb, ib = [], [], []
    for idx in shards:                      # what rank r would send: its sequences' records, containers renumbered locally
        it = torch.from_numpy(idx).to(dev)
        lo, hi = chs[it * per], chs[it * per + per]
        n = (hi - lo)
        loc_chs = torch.zeros(len(idx) * per + 1, dtype=torch.int64, device=dev)
        cnt = (chs[1:] - chs[:-1]).view(-1, per)[it].reshape(-1)
        torch.cumsum(cnt, 0, out=loc_chs[1:])
        pieces = torch.cat([hits[int(a):int(b)] for a, b in zip(lo.tolist(), hi.tolist())]) if len(idx) else hits[:0]
        h = pieces.clone()
        shift = torch.repeat_interleave(((torch.arange(len(idx), device=dev) - it) * per).to(torch.int32), n, output_size=h.shape[0])
        h[:, 0] += shift
        hb.append(h.reshape(-1).view(torch.uint8)); cb.append(loc_chs); ib.append(it)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        out, starts = kd.restore_hits(hb, cb, ib, len(lens), per)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ok = bool(torch.equal(out, hits)) and bool(torch.equal(starts, chs))
    print(json.dumps({"world": world, "hits": int(hits.shape[0]), "restore_ms": [round(x, 3) for x in ts], "identical_to_unsharded": ok}), flush=True)
    assert ok
