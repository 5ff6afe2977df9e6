"""Differential fuzz of the two scan strategies (tools/fuzz_strategies.py): random tables, sequences with planted
signatures and low-complexity runs, parameters, forced chunking, tiny regions and lists; every record kind and the event
bytes must be byte-identical between direct and partitioned probing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 4])
def test_strategies_agree_on_random_workloads(seed):
    env = {k: v for k, v in os.environ.items() if not k.startswith("KG_P") and k != "KG_TEST_TINY_LISTS"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_strategies.py"), "30", str(seed)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    last = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert last["all_identical"] and last["ran_partitioned"] >= 20
