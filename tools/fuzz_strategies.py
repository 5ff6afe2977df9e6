#!/usr/bin/env python3
"""Differential fuzz: direct vs partitioned strategy on random small workloads (tables, loads, DNA / protein, ragged
and low-complexity sequences, parameters, forced chunking, tiny regions, tiny lists).  Every record kind and the event
bytes must be byte-identical.  usage: fuzz_strategies.py [iterations] [seed]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import hotpath, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KNOBS = ("KG_PARTITION", "KG_PART_CHUNKS", "KG_PART_MIN_CHUNK_BLOCKS", "KG_PART_SLACK", "KG_TEST_TINY_LISTS", "KG_PART_OVF_GROUPS")
n_part = n_fallback = 0
for it in range(iters):
    aa = bool(rng.integers(0, 2))
    num_sigs = int(rng.choice([101, 1009, 50021, 200003, 1_000_003, 3_000_017]))
    load = float(rng.uniform(0.2, 0.95))
    n_keys = max(8, int(num_sigs * load))
    keys = synth.random_keys(n_keys, int(rng.integers(1, 1 << 30)))
    rec, placed = synth.build_table(keys, synth.payload_of(keys, int(rng.integers(1, 1 << 30)), n_otu=int(rng.integers(1, 9)), n_fn=int(rng.integers(1, 12))), num_sigs)
    img = synth.table_image(rec)
    n_seqs = int(rng.integers(1, 60))
    lens = rng.choice([0, 7, 23, 24, 64, 191, 192, 193, 500, 3000, 20000], size=n_seqs).astype(np.int64)
    if rng.integers(0, 3) == 0:
        lens[int(rng.integers(0, n_seqs))] = int(rng.integers(30000, 150000))
    off = np.zeros(n_seqs + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    raw = (synth.random_protein(max(total, 1), int(rng.integers(1, 1 << 30))) if aa else synth.random_dna(max(total, 1), int(rng.integers(1, 1 << 30)))).numpy()[:total].copy()
    kl = keys.tolist()
    # plant signatures and low-complexity runs
    for k in range(n_seqs):
        a, b = int(off[k]), int(off[k + 1])
        span = 8 if aa else 24
        p = a + int(rng.integers(0, 40))
        step = int(rng.integers(span, 120))
        while p + span <= b and rng.integers(0, 5) != 0:
            pep = synth.decode_kmer(int(kl[int(rng.integers(0, len(kl)))]))
            word = pep if aa else synth.back_translate(pep)
            raw[p:p + span] = np.frombuffer(word.encode(), dtype=np.uint8)
            p += step
        if b - a > 2000 and rng.integers(0, 3) == 0:
            q = a + int(rng.integers(0, b - a - 1500)); ln = int(rng.integers(300, 1500))
            unit = (b"K", b"KR", b"A")[int(rng.integers(0, 3))] if aa else (b"A", b"AT", b"ACG", b"T")[int(rng.integers(0, 4))]
            raw[q:q + ln] = np.frombuffer((unit * (ln // len(unit) + 1))[:ln], dtype=np.uint8)
        if b - a > 50 and rng.integers(0, 4) == 0:
            raw[a + int(rng.integers(0, b - a))] = ord("N") if not aa else ord("X")
    params = hotpath.Params(aa=aa, order_constraint=bool(rng.integers(0, 2)), min_hits=int(rng.integers(2, 7)),
                            min_weighted_hits=int(rng.integers(0, 4)), max_gap=int(rng.choice([5, 30, 200, 300])), counters=True)
    env = {"KG_PART_CHUNKS": str(int(rng.integers(1, 6))), "KG_PART_MIN_CHUNK_BLOCKS": "1"}
    if rng.integers(0, 3) == 0: env["KG_PART_SLACK"] = str(int(rng.choice([5, 20, 50])))
    if rng.integers(0, 4) == 0: env["KG_TEST_TINY_LISTS"] = "1"
    if rng.integers(0, 8) == 0: env["KG_PART_OVF_GROUPS"] = str(int(rng.choice([1, 64])))
    out = {}
    with hotpath.SignatureTable.from_bytes(img) as tab:
        for mode in ("0", "1"):
            for k in KNOBS: os.environ.pop(k, None)
            os.environ["KG_PARTITION"] = mode
            if mode == "1": os.environ.update(env)
            with tab.scan(raw, off, params) as r:
                st = r.stats
                out[mode] = (r.hits().tobytes(), r.container_hit_start().tobytes(), r.calls().tobytes(), r.container_call_start().tobytes(),
                             r.otu().tobytes(), r.hit_events().tobytes(), r.container_tail_events().tobytes(),
                             st["windows_valid"], st["slots_inspected"], st["residues"])
                if mode == "1":
                    n_part += st["partitioned"]; n_fallback += 1 - st["partitioned"]
    same = out["0"] == out["1"]
    print(json.dumps({"it": it, "aa": aa, "num_sigs": num_sigs, "load": round(load, 2), "n_seqs": n_seqs, "bp": total, "hits": len(out["0"][0]) // 24,
                      "calls": len(out["0"][2]) // 24, "env": env, "same": same}), flush=True)
    assert same, "strategies disagree"
for k in KNOBS: os.environ.pop(k, None)
print(json.dumps({"iterations": iters, "ran_partitioned": n_part, "fell_back_to_direct": n_fallback, "all_identical": True}))
