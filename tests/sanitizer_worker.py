#!/usr/bin/env python3
"""Child process of tests/test_sanitizers.py: the CPU oracle built with -fsanitize=address,undefined
(oracle/libkgoracle_asan.so, chosen through KGO_LIB_PATH; libasan preloaded by the parent) over the known-answer cases
and the fuzz workloads.  A sanitizer finding aborts the process; a wrong record fails an assertion.  Test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import kgo
    assert kgo.LIB_PATH.endswith("libkgoracle_asan.so"), kgo.LIB_PATH
    kgo.load()
    import kat_cases as K
    import test_oracle_kat as T
    n = 0
    for name in sorted(dir(T)):                                  # K1..K21 + the Java %f cases, on the sanitized build
        f = getattr(T, name)
        if name.startswith("test_") and callable(f):
            f(kgo)
            n += 1
    from fuzz_workloads import workloads
    rec = 0
    for w in workloads(int(sys.argv[1]) if len(sys.argv) > 1 else 12, 4242):
        p = w["params"]
        outs = [kgo.run(w["img"], w["raw"], w["off"], lookup_mode=mode, **p) for mode in (0, 1)]
        for k in ("hits", "calls", "otu", "hit_events", "container_tail_events"):
            assert outs[0][k].tobytes() == outs[1][k].tobytes(), (w["it"], k)        # merge-join == direct probing
        rec += len(outs[0]["hits"])
        # ragged edges: a truncated table image and an image with trailing records beyond numSigs
        img = bytes(w["img"])
        for cut in (img[:24 + 24 * 5 + 7], img + img[24:24 + 48]):
            try:
                kgo.run(cut, w["raw"], w["off"], lookup_mode=0, **p)
            except RuntimeError:
                pass
    print("sanitizer worker ok: %d test functions, %d fuzz hit records" % (n, rec))


if __name__ == "__main__":
    main()
