#!/usr/bin/env python3
"""Long differential fuzz against the CPU oracle (test infrastructure; not collected by pytest):
    python tests/fuzz_long.py [n_workloads=400] [first_seed=1000]
Every workload of tests/fuzz_workloads.py through the scan strategies (direct, behind the bit-per-slot digest and not; partitioned on the byte home index and on the tags; each with and without counters), records, event bytes, counters and flags compared
with the oracle.  Prints one JSON summary line; exit code 1 on the first difference."""
import json
import os
import sys

os.environ.setdefault("KG_ENABLE_TEST_HOOKS", "1")      # KG_TEST_TINY_LISTS workloads (include/kmerguts_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_workloads import workloads          # noqa: E402
from helpers import assert_same_records       # noqa: E402

KNOBS = ("KG_PARTITION", "KG_PART_CHUNKS", "KG_PART_MIN_CHUNK_BLOCKS", "KG_PART_SLACK", "KG_TEST_TINY_LISTS", "KG_PART_OVF_GROUPS",
         "KG_BIDX", "KG_INDEX_R", "KG_DIRECT_FILTER")


def main():
    from kmergutsjava_amd import hotpath
    from oracle import kgo
    kgo.build()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    done = n_part = calls = hits = 0
    for seed in range(seed0, seed0 + (n + 24) // 25):
        for w in workloads(25, seed):
            p = w["params"]
            ora = kgo.run(w["img"], w["raw"], w["off"], lookup_mode=1, **p)
            with hotpath.SignatureTable.from_bytes(w["img"]) as tab:
                for mode in ("0", "1", "2"):           # direct, partitioned (byte home index / tags with counters), partitioned on the tags only
                    for k in KNOBS:
                        os.environ.pop(k, None)
                    os.environ["KG_PARTITION"] = "0" if mode == "0" else "1"
                    os.environ["KG_DIRECT_FILTER"] = "2" if w["it"] % 2 == 0 else "1"      # (2: the bit-per-slot digest whatever the table's size)
                    if mode != "0":
                        os.environ.update(w["env"])
                    if mode == "2":
                        os.environ.update(w["env2"])
                    with tab.scan(w["raw"], w["off"], hotpath.Params(counters=True, **p)) as r:
                        assert_same_records(r, ora, "fuzz seed %d it %d mode %s %s %s" % (seed, w["it"], mode, w["env"], w["env2"] if mode == "2" else ""))
                        assert r.stats["windows_valid"] == ora["windows_valid"] and r.stats["slots_inspected"] == ora["slots_inspected"]
                        n_part += r.stats["partitioned"]
                    if True:                               # without KG_F_COUNTERS: the direct kernel behind the digest ("0") / the byte-index kernel ("1") / the plain tag kernel ("2")
                        with tab.scan(w["raw"], w["off"], hotpath.Params(**p)) as r:
                            assert_same_records(r, ora, "fuzz seed %d it %d mode %s no counters %s" % (seed, w["it"], mode, w["env"]))
            done += 1
            if done % 100 == 0:                                        # (a run that writes nothing for minutes is taken to be hung)
                print("[fuzz_long] %d workloads identical so far" % done, file=sys.stderr, flush=True)
            calls += len(ora["calls"])
            hits += len(ora["hits"])
            if done >= n:
                break
    print(json.dumps({"workloads": done, "scans": 6 * done, "ran_partitioned": n_part, "oracle_hits": hits, "oracle_calls": calls,
                      "all_identical": True}))


if __name__ == "__main__":
    main()
