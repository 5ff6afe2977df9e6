"""The hand-derived known-answer cases K16..K21 (tests/kat_cases.py) on the HIP path, through the C ABI:
kg_aggregate_hits for the gatherHits / processSetOfHits cases, kg_scan (every scan strategy) for the lookup cases."""
import numpy as np
import pytest

import kat_cases as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make", K.AGGREGATION_CASES, ids=lambda f: f.__name__)
def test_aggregation_cases_through_kg_aggregate_hits(make):
    from kmergutsjava_amd import hotpath
    for name, h, kw, want_calls, want_otu in make():
        p = hotpath.Params(aa=True, **kw)                  # one container per sequence
        with hotpath.aggregate_hits(h, [0, len(h)], 1, p) as r:
            calls = [(int(c["start"]), int(c["end"]), int(c["count"]), int(c["fI"]), float(c["weightedHits"])) for c in r.calls()]
            o = r.otu()[0]
            otu = [(int(o["count"][j]), int(o["oI"][j])) for j in range(int(o["n"]))]
        assert calls == want_calls, (name, calls)
        assert otu == want_otu, (name, otu)


@pytest.mark.parametrize("strategy", ["direct", "partitioned", "partitioned2"])
@pytest.mark.parametrize("make", K.LOOKUP_CASES, ids=lambda f: f.__name__)
def test_lookup_cases_through_kg_scan(make, strategy, monkeypatch):
    from kmergutsjava_amd import hotpath
    monkeypatch.setenv("KG_PARTITION", "0" if strategy == "direct" else "1")
    monkeypatch.setenv("KG_PART_LEVELS", "2" if strategy == "partitioned2" else "1")
    for name, img, q, want in make():
        with hotpath.SignatureTable.from_bytes(img) as tab, tab.scan(q, np.array([0, len(q)]), hotpath.Params(aa=True, min_hits=2)) as r:
            got = [(int(h["from0InProt"]), int(h["oI"]), int(h["avgOffFromEnd"]), int(h["fI"]), float(h["functionWt"]))
                   for h in r.hits()]
            assert got == want, (name, strategy, got)
            if strategy != "direct":
                assert r.stats["partitioned"] == 1, name
