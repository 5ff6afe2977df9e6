"""Differential fuzz (tests/fuzz_workloads.py: random tables, sequences with planted signatures and low-complexity
runs, parameters, forced chunking, tiny regions and lists):
  * the scan strategies (direct; partitioned on the byte home index and on the tags) against the CPU oracle, record for record, events included;
  * tools/fuzz_strategies.py (direct vs partitioned only, no oracle) as a subprocess."""
import json
import os
import subprocess
import sys

import pytest

from fuzz_workloads import workloads
from helpers import assert_same_records

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ("KG_PARTITION", "KG_PART_CHUNKS", "KG_PART_MIN_CHUNK_BLOCKS", "KG_PART_SLACK", "KG_TEST_TINY_LISTS", "KG_PART_OVF_GROUPS",
         "KG_BIDX", "KG_INDEX_R", "KG_DIRECT_FILTER")


@pytest.mark.parametrize("seed", [21, 22])
def test_both_strategies_against_the_oracle(oracle, monkeypatch, seed):
    from kmergutsjava_amd import hotpath
    n_part = n_bidx = n_tags = 0
    for w in workloads(25, seed):
        p = w["params"]
        ora = oracle.run(w["img"], w["raw"], w["off"], lookup_mode=1, **p)
        with hotpath.SignatureTable.from_bytes(w["img"]) as tab:
            for mode in ("0", "1", "2"):               # direct, partitioned (byte home index / tags with counters), partitioned on the tags only
                for k in KNOBS:
                    monkeypatch.delenv(k, raising=False)
                monkeypatch.setenv("KG_PARTITION", "0" if mode == "0" else "1")
                monkeypatch.setenv("KG_DIRECT_FILTER", "2" if w["it"] % 2 == 0 else "1")      # (2: the digest whatever the table's size)
                if mode != "0":
                    for k, v in w["env"].items():
                        monkeypatch.setenv(k, v)
                if mode == "2":
                    for k, v in w["env2"].items():
                        monkeypatch.setenv(k, v)
                with tab.scan(w["raw"], w["off"], hotpath.Params(counters=True, **p)) as r:
                    assert_same_records(r, ora, "fuzz seed %d it %d mode %s %s %s" % (seed, w["it"], mode, w["env"], w["env2"] if mode == "2" else ""))
                    assert r.stats["windows_valid"] == ora["windows_valid"] and r.stats["slots_inspected"] == ora["slots_inspected"]
                    n_part += r.stats["partitioned"]
                if mode == "0":
                    # without KG_F_COUNTERS the direct kernel asks the table's bit-per-slot digest before the tags
                    with tab.scan(w["raw"], w["off"], hotpath.Params(**p)) as r:
                        assert_same_records(r, ora, "fuzz seed %d it %d direct, no counters" % (seed, w["it"]))
                if mode != "0":
                    # without KG_F_COUNTERS: mode "1" probes the table's byte home index (bucket_index_kernel), mode "2" the tags
                    with tab.scan(w["raw"], w["off"], hotpath.Params(**p)) as r:
                        assert_same_records(r, ora, "fuzz seed %d it %d mode %s no counters %s" % (seed, w["it"], mode, w["env"]))
                        n_bidx += r.stats["part_levels"] == 4
                        n_tags += r.stats["part_levels"] == 1
    assert n_part >= 30 and n_bidx >= 10 and n_tags >= 10


def test_strategies_agree_on_random_workloads():
    env = {k: v for k, v in os.environ.items() if k not in KNOBS}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_strategies.py"), "30", "3"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    last = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert last["all_identical"] and last["ran_partitioned"] >= 20
