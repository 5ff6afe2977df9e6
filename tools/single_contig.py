#!/usr/bin/env python3
"""ONE long contig against the full-size table: stage times (SW_BP, default 300 Mbp), and with SW_CHECK_BP > 0 a parity run of
a shorter single contig against the oracle (both strategies).  Tuning / sanity aid for the one-sequence shape: one chunk in the
partitioned pipeline, six long containers in the aggregation."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmergutsjava_amd import hotpath, synth
num_sigs = int(os.environ.get("SW_NUM_SIGS", "1400303159"))
bp = int(os.environ.get("SW_BP", "300000000")); check_bp = int(os.environ.get("SW_CHECK_BP", "0"))
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev); del keys
torch.cuda.synchronize()
tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
off = np.array([0, bp], dtype=np.int64)
seq = synth.random_dna(bp, 311, dev); torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
        r.calls(); st = r.stats
    wall = (time.perf_counter() - t0) * 1e3
print(json.dumps({"bp": bp, "wall_ms": wall, **{k: st[k] for k in ("ms_scan", "ms_order", "ms_aggregate", "n_hits", "n_calls", "partitioned", "part_chunks", "agg_pieces")}}), flush=True)
if check_bp:
    import struct
    from oracle import kgo
    from helpers import assert_same_records
    kgo.build()
    host = torch.empty(24 + num_sigs * 24, dtype=torch.uint8)
    host[:24] = torch.frombuffer(bytearray(struct.pack("<qqq", num_sigs, 24, 1)), dtype=torch.uint8)
    host[24:].view(torch.int32).view(num_sigs, 6).copy_(rec)
    off2 = np.array([0, check_bp], dtype=np.int64)
    sub = seq[:check_bp].cpu().numpy()
    ora = kgo.run(host.numpy(), sub, off2, lookup_mode=1)
    for mode in ("0", "1"):
        os.environ["KG_PARTITION"] = mode
        with tab.scan(None, off2, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
            assert_same_records(r, ora, "single contig mode " + mode)
            print(json.dumps({"check_bp": check_bp, "mode": mode, "identical": True, "n_hits": r.stats["n_hits"], "agg_pieces": r.stats["agg_pieces"], "partitioned": r.stats["partitioned"]}), flush=True)
