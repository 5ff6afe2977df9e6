#!/usr/bin/env python3
"""Generates tests/golden/vectors_r01.json.

PROVENANCE: the reference (Java) cannot run in the build image and holds no input->output pair for
this path, so these vectors are NOT reference outputs.  They are the outputs of the independent
pure-Python restatement oracle/kgj_model.py (literal merge-join lookup, Java-exact text) on small
deterministic inputs, frozen so that the C oracle, the HIP path and the host text layer are all
checked against the same bytes in every round.  Parity status: unpinned (see oracle/kg_oracle.h).

The Ecoli_K12_W3110.{fna,faa}.gz files next to this script are the reference's own test inputs
(test/data/ of the reference; data, not source); no expected output exists for them.
"""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from kmergutsjava_amd import synth          # noqa: E402  (input generators only)
from oracle import kgj_model as M           # noqa: E402


def fasta_of(sb, off, width=70):
    out = []
    for k in range(len(off) - 1):
        s = sb[off[k]:off[k + 1]].decode()
        out.append(">g%d golden vector %d\n" % (k, k))
        out += [s[i:i + width] + "\n" for i in range(0, len(s), width)]
    return "".join(out)


def main():
    vectors = []
    fn = ["golden function %d" % i for i in range(64)]
    for name, dna, seed, params in (
        ("dna_default", True, 901, dict()),
        ("dna_m2_g8_M3", True, 902, dict(min_hits=2, max_gap=8, min_weighted_hits=3)),
        ("dna_order_constraint", True, 903, dict(order_constraint=True, min_hits=3, max_gap=30)),
        ("aa_default", False, 904, dict()),
        ("aa_order_constraint", False, 905, dict(order_constraint=True, min_hits=2, max_gap=8)),
    ):
        seq, off, rec, keys = synth.high_density_config(4, 40, 1009, 400, seed=seed, dna=dna)
        img = synth.table_image(rec)
        sb = seq.numpy().tobytes()
        m = M.Model(aa=not dna, **params)
        fa = fasta_of(sb, off)
        text = m.run(img, fn, fa)
        vectors.append({
            "name": name, "aa": not dna, "params": params,
            "table_b64": base64.b64encode(img).decode(), "fasta": fa, "functions": fn,
            "hits": [list(map(float, h)) for h in m.hits],
            "calls": [list(map(float, c)) for c in m.calls],
            "otu": m.otus, "report": text,
        })
        print(name, len(m.hits), "hits", len(m.calls), "calls", file=sys.stderr)
    json.dump({"provenance": "oracle/kgj_model.py (pure-Python restatement); NOT reference output", "vectors": vectors},
              open(os.path.join(HERE, "vectors_r01.json"), "w"))


if __name__ == "__main__":
    main()
