"""The public instance methods of the reference class (processSetOfHits KGJ:385, gatherHits KGJ:457,
processAASeq KGJ:526, createKmerComparator KGJ:1082) on the host mirror, which runs them on the GPU through
kg_process_set_of_hits / kg_aggregate_hits, against the literal pure-Python model of the reference."""
import functools
import io

import numpy as np
import pytest

from oracle import kgj_model as M


def _random_hits(rng, n, n_fi=4, n_oi=7, spread=60):
    pos = np.sort(rng.choice(np.arange(n * spread // 4 + n), size=n, replace=False))
    return [(int(rng.integers(0, n_oi)), int(p), int(rng.integers(0, 40)), int(rng.integers(0, n_fi)),
             float(rng.integers(1, 65)) / 16.0) for p in pos]


def test_comparator_and_types():
    from kmergutsjava_amd import KmerGutsJava as K
    cmp = K.createKmerComparator(1009)
    rng = np.random.default_rng(1)
    qs = [K.QueryKmer(int(v), i, i) for i, v in enumerate(rng.integers(0, 20 ** 8, size=500))]
    got = sorted(qs, key=functools.cmp_to_key(cmp))
    want = sorted(qs, key=lambda q: (q.value % 1009, q.value))
    assert [q.value for q in got] == [q.value for q in want]
    a, b = K.HitContainerKey("x", "+", 1), K.HitContainerKey("x", "+", 1)
    assert a == b and hash(a) == hash(b) and a != K.HitContainerKey("x", "-", 1) and {a: 1}[b] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("debug", [False, True])
def test_gather_hits_and_process_aa_seq(debug):
    from kmergutsjava_amd import KmerGutsJava as K
    fn = ["function %d" % i for i in range(8)]
    rng = np.random.default_rng(11)
    for trial in range(12):
        oc = bool(trial & 1)
        mh, gap, mw = ((5, 200, 0), (2, 15, 0), (3, 40, 2))[trial % 3]
        tup = _random_hits(rng, int(rng.integers(2, 120)), n_fi=2 + trial % 3)
        rng.shuffle(tup)                                                     # gatherHits sorts its argument
        model = M.Model(aa=True, order_constraint=oc, min_hits=mh, min_weighted_hits=mw, max_gap=gap, debug=debug)
        k = K()
        k.aa, k.orderConstraint, k.minHits, k.minWeightedHits, k.maxGap, k.debug = True, oc, mh, mw, gap, debug
        start = [(3, 2), (2, 5)] if trial % 4 == 0 else []                   # (count, oI): a buffer that is not empty
        if trial % 4 == 2:       # ... and one no run of the reference could have left behind: an otuIndex twice, counts out of order
            start = [(1, 2), (4, 5), (2, 2), (6, 3)]
        m_hits = [M.Hit(*t) for t in tup]
        m_oi = [[c, o] for c, o in start]
        m_pw = io.StringIO()
        model.gather_hits(500, "+", 0, m_hits, fn, m_oi, m_pw)
        g_hits = [K.Hit(*t) for t in tup]
        g_oi = [K.OtuCount(o, c) for c, o in start]
        g_pw = io.StringIO()
        k.gatherHits(500, "+", 0, g_hits, fn, g_oi, g_pw)
        assert g_pw.getvalue() == m_pw.getvalue(), trial
        assert [(x.count, x.oI) for x in g_oi] == [(c, o) for c, o in m_oi], trial
        assert [h.from0InProt for h in g_hits] == [h.from0InProt for h in m_hits]         # sorted in place
        # processAASeq: header, CALLs, OTU-COUNTS of one protein
        m_pw, g_pw = io.StringIO(), io.StringIO()
        model.process_aa_seq("p1", 321, {("p1", "+", 0): {"id": 0, "hits": [M.Hit(*t) for t in tup]}}, fn, m_pw)
        k.processAASeq("p1", 321, {K.HitContainerKey("p1", "+", 0): K.HitContainer(None, 0, [K.Hit(*t) for t in tup])}, fn, g_pw)
        assert g_pw.getvalue() == m_pw.getvalue(), trial


@pytest.mark.gpu
def test_aggregate_hits_result_views():
    """kg_aggregate_hits hands out an ordinary result: host views, device views and kg_result_copy_hits work on it."""
    from kmergutsjava_amd import hotpath, _native as N
    rng = np.random.default_rng(3)
    tup = sorted(_random_hits(rng, 300), key=lambda t: t[1])
    rec = np.zeros(len(tup), dtype=N.HIT_DTYPE)
    for i, (o, pos, avg, fi, wt) in enumerate(tup):
        rec[i] = (0, pos, o, avg, fi, wt)
    with hotpath.aggregate_hits(rec, [0, len(rec)], 1, hotpath.Params(aa=True, min_hits=2)) as r:
        assert r.stats["n_hits"] == len(rec) and r.stats["n_calls"] == len(r.calls()) > 0
        assert r.hits().tobytes() == rec.tobytes() == r.copy_hits().tobytes()
        assert r.device_view("hits").cpu().numpy().tobytes() == rec.tobytes()
        assert int(r.container_call_start()[-1]) == r.stats["n_calls"] and len(r.hit_events()) == len(rec)


@pytest.mark.gpu
def test_process_set_of_hits():
    from kmergutsjava_amd import KmerGutsJava as K
    fn = ["function %d" % i for i in range(8)]
    rng = np.random.default_rng(5)
    n_called = n_keep = 0
    for trial in range(40):
        mh, mw = ((5, 0), (2, 0), (3, 4))[trial % 3]
        tup = _random_hits(rng, int(rng.integers(2, 30)), n_fi=2 + trial % 2)
        cur = int(rng.integers(0, 3))
        start = [(4, 1), (1, 6), (1, 3), (1, 2), (1, 0)] if trial % 5 == 0 else ([(2, 3)] if trial % 2 else [])
        for debug in (False, True):
            model = M.Model(aa=True, min_hits=mh, min_weighted_hits=mw, debug=debug)
            k = K()
            k.minHits, k.minWeightedHits, k.debug = mh, mw, debug
            m_hits, m_oi, m_pw = [M.Hit(*t) for t in tup], [[c, o] for c, o in start], io.StringIO()
            g_hits, g_oi, g_pw = [K.Hit(*t) for t in tup], [K.OtuCount(o, c) for c, o in start], io.StringIO()
            want = model.process_set_of_hits(m_hits, fn, cur, m_oi, m_pw)
            got = k.processSetOfHits(g_hits, fn, cur, g_oi, g_pw)
            assert got == want and g_pw.getvalue() == m_pw.getvalue(), trial
            assert [(h.from0InProt, h.fI) for h in g_hits] == [(h.from0InProt, h.fI) for h in m_hits]
            assert [(x.count, x.oI) for x in g_oi] == [(c, o) for c, o in m_oi]
        n_called += "CALL" in m_pw.getvalue()
        n_keep += len(m_hits) == 2
    assert n_called >= 5 and n_keep >= 5
    with pytest.raises(IndexError):
        K().processSetOfHits([K.Hit()], fn, 0, [], io.StringIO())            # hits.get(numHits - 2) throws in the reference
