#!/usr/bin/env python3
"""Differential fuzz: direct vs partitioned (tags, byte home index) strategy on random small workloads (tests/fuzz_workloads.py: tables, loads,
DNA / protein, ragged and low-complexity sequences, parameters, forced chunking, tiny regions, tiny lists).  Every
record kind and the event bytes must be byte-identical.  usage: fuzz_strategies.py [iterations] [seed]"""
import json, os, sys
os.environ.setdefault("KG_ENABLE_TEST_HOOKS", "1")      # KG_TEST_TINY_LISTS workloads (include/kmerguts_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmergutsjava_amd import hotpath
from fuzz_workloads import workloads

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KNOBS = ("KG_PARTITION", "KG_PART_CHUNKS", "KG_PART_MIN_CHUNK_BLOCKS", "KG_PART_SLACK", "KG_TEST_TINY_LISTS", "KG_PART_OVF_GROUPS",
         "KG_BIDX", "KG_INDEX_R")
n_part = n_fallback = 0
for w in workloads(iters, seed):
    out = {}
    with hotpath.SignatureTable.from_bytes(w["img"]) as tab:
        for mode in ("0", "1", "2"):               # direct, partitioned (byte home index / tags with counters), partitioned on the tags only
            for k in KNOBS: os.environ.pop(k, None)
            os.environ["KG_PARTITION"] = "0" if mode == "0" else "1"
            if mode != "0": os.environ.update(w["env"])
            if mode == "2": os.environ.update(w["env2"])
            with tab.scan(w["raw"], w["off"], hotpath.Params(counters=True, **w["params"])) as r:
                st = r.stats
                out[mode] = (r.hits().tobytes(), r.container_hit_start().tobytes(), r.calls().tobytes(), r.container_call_start().tobytes(),
                             r.otu().tobytes(), r.hit_events().tobytes(), r.container_tail_events().tobytes(),
                             st["windows_valid"], st["slots_inspected"], st["residues"])
                if mode == "1":
                    n_part += st["partitioned"]; n_fallback += 1 - st["partitioned"]
            if mode == "1":                                # ... and through the byte home index (no KG_F_COUNTERS kernel)
                with tab.scan(w["raw"], w["off"], hotpath.Params(**w["params"])) as r:
                    out["3"] = (r.hits().tobytes(), r.container_hit_start().tobytes(), r.calls().tobytes(), r.container_call_start().tobytes(),
                                r.otu().tobytes(), r.hit_events().tobytes(), r.container_tail_events().tobytes()) + out["1"][7:]
    same = out["0"] == out["1"] == out["2"] == out["3"]
    print(json.dumps({"it": w["it"], "aa": w["aa"], "num_sigs": w["num_sigs"], "load": round(w["load"], 2), "n_seqs": len(w["off"]) - 1,
                      "bp": int(w["off"][-1]), "hits": len(out["0"][0]) // 24, "calls": len(out["0"][2]) // 24, "env": w["env"], "env2": w["env2"], "same": same}), flush=True)
    assert same, "strategies disagree"
for k in KNOBS: os.environ.pop(k, None)
print(json.dumps({"iterations": iters, "ran_partitioned": n_part, "fell_back_to_direct": n_fallback, "all_identical": True}))
