"""The hand-derived known-answer cases on the HIP path, through the C ABI: K1..K15 of SURVEY.md section 8c (expected values
copied from that table -- NOT computed by the oracle or by any helper that shares code with it) and K16..K21 of
tests/kat_cases.py; kg_aggregate_hits for the gatherHits / processSetOfHits cases, kg_scan (every scan strategy) for the
encode / translate / window / lookup cases.  With parity unpinned (no JVM, no reference fixture) these are the only
anchors that do not pass through one of the repo's own restatements."""
import struct

import numpy as np
import pytest

import kat_cases as K

A, B = 7, 9                                                  # two function indices
EMPTY = 20 ** 8 + 1                                          # whichKmer > MAX_ENCODED: empty slot (KGJ:1000)
STRATEGIES = ["direct", "partitioned", "partitioned_tags"]
# one codon per residue, standard code (KGJ:88-93), written out here so that the DNA cases do not lean on synth.back_translate
CODON = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTG",
         "M": "ATG", "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}
# K1: 8-mer -> encodedKmer (KGJ:274-292, alphabet order ACDEFGHIKLMNPQRSTVWY of KGJ:111-175)
K1 = {"AAAAAAAA": 0, "ACDEFGHI": 70914127, "MKLVTGAS": 13343650015, "YYYYYYYY": 25599999999}


def _strategy(monkeypatch, strategy):
    monkeypatch.setenv("KG_PARTITION", "0" if strategy == "direct" else "1")
    monkeypatch.setenv("KG_DIRECT_FILTER", "2")          # the direct kernel behind the bit-per-slot digest, whatever the table's size
    monkeypatch.setenv("KG_BIDX", "0" if strategy == "partitioned_tags" else "1")


def _image(n, entries, extra=b""):
    """F1 table image (KGJ:924-942, 995-999): header {numSigs, entrySize 24, version}, then n records
    {int64 whichKmer, int32 otuIndex, int32 avgFromEnd, int32 functionIndex, float32 functionWt}, little endian."""
    body = bytearray()
    for i in range(n):
        k, o, a, f, w = entries.get(i, (EMPTY, 0, 0, 0, 0.0))
        body += struct.pack("<qiiif", k, o, a, f, w)
    return struct.pack("<qqq", n, 24, 1) + bytes(body) + extra


def _table_of(values, n=1009):
    """Every value at its home slot value % n (the values of these cases do not collide); payload oI = index + 1."""
    ent = {}
    for i, v in enumerate(values):
        assert v % n not in ent
        ent[v % n] = (v, i + 1, 10 * (i + 1), 100 + i, 0.5 * (i + 1))
    return _image(n, ent)


def _scan_hits(img, q, aa, counters=False, min_hits=2):
    from kmergutsjava_amd import hotpath
    q = q if isinstance(q, bytes) else q.encode()
    with hotpath.SignatureTable.from_bytes(img) as tab, \
            tab.scan(q, np.array([0, len(q)]), hotpath.Params(aa=aa, min_hits=min_hits, counters=counters)) as r:
        return [(int(h["container"]), int(h["from0InProt"]), int(h["oI"])) for h in r.hits()], dict(r.stats), \
               [(int(c["container"]), int(c["start"]), int(c["end"]), int(c["count"]), int(c["fI"])) for c in r.calls()], r.otu().copy()


def _agg(h, **kw):
    from kmergutsjava_amd import hotpath
    p = hotpath.Params(aa=True, **kw)                        # one container per sequence
    with hotpath.aggregate_hits(h, [0, len(h)], 1, p) as r:
        calls = [(int(c["start"]), int(c["end"]), int(c["count"]), int(c["fI"]), float(c["weightedHits"])) for c in r.calls()]
        o = r.otu()[0]
        return calls, [(int(o["count"][j]), int(o["oI"][j])) for j in range(int(o["n"]))]


def _hits(fis, pos=None, ois=None, wts=None):
    n = len(fis)
    h = np.zeros(n, dtype=K.HIT)
    h["from0InProt"] = list(range(n)) if pos is None else pos
    h["oI"] = [3] * n if ois is None else ois
    h["fI"] = fis
    h["functionWt"] = [1.0] * n if wts is None else wts
    return h


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K1_encoded_kmer_through_kg_scan(strategy, monkeypatch):
    """The table holds exactly the four K1 values; a hit says the kernel's window value equalled the stored int64
    (the lookup compares whichKmer with the query's value, KGJ:1003).  Protein AND DNA (back-translated) input."""
    _strategy(monkeypatch, strategy)
    img = _table_of(list(K1.values()))
    order = list(K1)
    # proteins: each 8-mer followed by one more residue (AA mode never queries the last window, KGJ:912), separated by X
    q = "X".join(k + "A" for k in order)
    got, st, _, _ = _scan_hits(img, q, aa=True, counters=True)
    # windows: per 9-residue piece start 0 is the K1 8-mer; start 1 of "AAAAAAAA"+"A" is AAAAAAAA again (value 0)
    assert got == [(0, 0, 1), (0, 1, 1), (0, 10, 2), (0, 20, 3), (0, 30, 4)], got
    dna = "TAA".join("".join(CODON[c] for c in k) for k in order)              # stop codons between the 24-mers
    got, st, _, _ = _scan_hits(img, dna, aa=False)
    assert got == [(0, 0, 1), (0, 9, 2), (0, 18, 3), (0, 27, 4)], got              # frame +0 (container 0) only


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K2_invalid_residues_are_never_queried(strategy, monkeypatch):
    """*, X, x, lowercase, U -> code 20 -> encodedKmer -1 -> no QueryKmer (KGJ:913-920): windows_valid counts only clean windows."""
    _strategy(monkeypatch, strategy)
    img = _table_of(list(K1.values()))
    for bad in ("AAAA*AAAA", "AAAAXAAAA", "AAAAxAAAA", "aaaaaaaaa", "AAAAAAAUA"):
        got, st, _, _ = _scan_hits(img, bad, aa=True, counters=True)
        assert got == [] and st["windows_valid"] == 0, (bad, got, st["windows_valid"])
    got, st, _, _ = _scan_hits(img, "AAAAAAAAA*AAAAAAAAAA", aa=True, counters=True)
    # windows 0 (AAAAAAAA) valid, 1 (AAAAAAAA) valid, 2..9 contain '*', 10 and 11 valid (the last, 12, is never queried)
    assert st["windows_valid"] == 4 and got == [(0, 0, 1), (0, 1, 1), (0, 10, 1), (0, 11, 1)], (got, st["windows_valid"])


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K3_translate_frames_and_stops(strategy, monkeypatch):
    """K3: ATG GCC TAA -> M A * in frame 0, TGG CCT -> W P in frame 1 (KGJ:294-343).  Long enough for windows here:
    frame 0 = MKLVTGAS then a stop; frame 1 of the same bases shifted by one = ACDEFGHI."""
    _strategy(monkeypatch, strategy)
    img = _table_of(list(K1.values()))
    f0 = "".join(CODON[c] for c in "MKLVTGAS") + "TAA"
    got, _, _, _ = _scan_hits(img, f0, aa=False)
    assert got == [(0, 0, 3)], got                                               # MKLVTGAS* : one clean window, frame +0
    f1 = "G" + "".join(CODON[c] for c in "ACDEFGHI") + "TA"
    got, _, _, _ = _scan_hits(img, f1, aa=False)
    assert got == [(1, 0, 2)], got                                               # container 1 = ('+', frame 1)
    # lowercase bases and u/U are bases (KGJ:294-318); any other character makes its codon 'x'
    low = "".join(CODON[c] for c in "MKLVTGAS").lower().replace("t", "u")
    got, _, _, _ = _scan_hits(img, low, aa=False)
    assert got == [(0, 0, 3)], got
    got, _, _, _ = _scan_hits(img, "".join(CODON[c] for c in "MKLV") + "GGN" + "".join(CODON[c] for c in "GAS"), aa=False)
    assert got == [], got
    # reverse strand: the reverse complement of the DNA hits in container 3 = ('-', frame 0), position 0
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[b] for b in reversed("".join(CODON[c] for c in "MKLVTGAS")))
    got, _, _, _ = _scan_hits(img, rc, aa=False)
    assert got == [(3, 0, 3)], got


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K4_window_counts(strategy, monkeypatch):
    """DNA of length L: frame off has floor((L - off) / 3) residues and windows i in [0, n_off - 8]; a protein of length n
    windows i in [0, n - 9] (KGJ:912, 1051-1074)."""
    _strategy(monkeypatch, strategy)
    img = _table_of(list(K1.values()))
    want_dna = {23: 0, 24: 2, 25: 4, 26: 6, 27: 8, 50: 54}                       # both strands, three frames, by hand
    for L, want in want_dna.items():
        _, st, _, _ = _scan_hits(img, ("GCT" * 40)[:L], aa=False, counters=True)
        assert st["windows_valid"] == want, (L, st["windows_valid"])
        assert st["residues"] == sum(2 * ((L - off) // 3) for off in range(3))
    for n, want in {8: 0, 9: 1, 10: 2, 30: 22}.items():
        _, st, _, _ = _scan_hits(img, "A" * n, aa=True, counters=True)
        assert st["windows_valid"] == want and st["residues"] == n, (n, st["windows_valid"])


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K5_no_wrap_and_probe_stop(strategy, monkeypatch):
    """11 slots.  Key 10 (home 10, the last slot) stored at slot 0 "after a wrap": NOT found; key 22 (home 0) stored at slot 1
    behind occupied slot 0: found; key 21 (home 10) in place: found.  Probing stops at the first empty slot (KGJ:944-1034)."""
    _strategy(monkeypatch, strategy)
    img = _image(11, {10: (21, 1, 0, 1, 1.0), 0: (10, 2, 0, 2, 1.0), 1: (22, 3, 0, 3, 1.0)})
    # 10 = AAAAAAAM (M = code 10), 22 = AAAAAACD (1 * 20 + 2), 21 = AAAAAACC (1 * 20 + 1): base-20 digits by hand
    q = "AAAAAAAM" + "A" + "AAAAAACD" + "A" + "AAAAAACC" + "A"
    got, _, _, _ = _scan_hits(img, q, aa=True)
    pos = {(p, o) for _, p, o in got}
    assert (9, 3) in pos and (18, 1) in pos and not any(p == 0 for p, _ in pos), got
    img2 = _image(11, {3: (25, 1, 0, 1, 1.0), 5: (3, 9, 0, 9, 1.0)})             # key 3 (home 3) sits behind a hole at slot 4
    got, _, _, _ = _scan_hits(img2, "AAAAAAAE" + "AA", aa=True)                   # 3 = AAAAAAAE (E = code 3)
    assert got == [], got


def test_K6_to_K14_through_kg_aggregate_hits():
    """Expected values: SURVEY.md 8c, rows K6..K14 (KGJ:385-514)."""
    assert _agg(_hits([A] * 5, pos=[0, 10, 20, 30, 40])) == ([(0, 47, 5, A, 5.0)], [(5, 3)])                     # K6
    assert _agg(_hits([A] * 4, pos=[0, 10, 20, 30])) == ([], [])
    assert _agg(_hits([A, A, A, A, A, B, B]))[0] == [(0, 11, 5, A, 5.0)]                                          # K7
    assert _agg(_hits([A, A, B, A, A, A]))[0] == [(0, 12, 5, A, 5.0)]                                             # K8
    assert _agg(_hits([B, A, A, A, A, A]))[0] == [(1, 12, 5, A, 5.0)]                                             # K9
    assert _agg(_hits([A, A, B, B, A, A, A, A, A]))[0] == [(4, 15, 5, A, 5.0)]                                    # K10
    first = [0, 10, 20, 30, 40]
    assert _agg(_hits([A] * 10, pos=first + [240, 250, 260, 270, 280]))[0] == [(0, 287, 10, A, 10.0)]             # K11
    assert _agg(_hits([A] * 10, pos=first + [241, 251, 261, 271, 281]))[0] == [(0, 47, 5, A, 5.0), (241, 288, 5, A, 5.0)]
    assert _agg(_hits([A] * 6, ois=[10, 11, 12, 13, 14, 15]))[1] == [(1, 15), (1, 14), (1, 13), (1, 12), (1, 11)]  # K12
    assert _agg(_hits([A] * 5, ois=[1, 2, 1, 2, 3]))[1] == [(2, 2), (2, 1), (1, 3)]                               # K13
    assert _agg(_hits([A] * 5, wts=[16777216.0, 1, 1, 1, 1]))[0][0][4] == 16777216.0                              # K14
    assert _agg(_hits([A] * 5, wts=[1, 1, 1, 1, 16777216.0]))[0][0][4] == 16777220.0


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_K15_otu_buffer_persists_across_frames(strategy, monkeypatch):
    """One contig: five signature 8-mers (oI 3, fI 7) back to back in frame +0, then five (oI 4, fI 8) on the reverse strand in
    frame 1 -> CALLs in containers 0 and 4, OTU-COUNTS 5-4 5-3: the buffer is shared by the six frames and a later equal count
    moves ahead (KGJ:432-437, 540-557)."""
    _strategy(monkeypatch, strategy)
    plus = ["ACDEFGHI", "MKLVTGAS", "WYWYWYWA", "CCDDEEFF", "HIKLHIKL"]
    minus = ["SAGTVLKM", "QQPPNNMM", "RSTVRSTV", "YAYAYAYC", "GHGHGHGD"]

    def val(s):
        v = 0
        for c in s:
            v = v * 20 + "ACDEFGHIKLMNPQRSTVWY".index(c)
        return v
    n = 5003
    ent = {}
    for i, s in enumerate(plus + minus):
        v = val(s)
        assert v % n not in ent
        ent[v % n] = (v, 3 if i < 5 else 4, 0, 7 if i < 5 else 8, 1.0)
    img = _image(n, ent)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    minus_src = "".join(CODON[c] for s in minus for c in s)
    contig = "".join(CODON[c] for s in plus for c in s) + "N" * 30 + "".join(comp[b] for b in reversed(minus_src)) + "G"
    got, _, calls, otu = _scan_hits(img, contig, aa=False, min_hits=5)
    assert sorted({c[0] for c in calls}) == [0, 4], calls
    assert [(c[3], c[4]) for c in calls] == [(5, 7), (5, 8)], calls
    o = otu[0]
    assert [(int(o["count"][j]), int(o["oI"][j])) for j in range(int(o["n"]))] == [(5, 4), (5, 3)]

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


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("make", K.LOOKUP_CASES, ids=lambda f: f.__name__)
def test_lookup_cases_through_kg_scan(make, strategy, monkeypatch):
    from kmergutsjava_amd import hotpath
    _strategy(monkeypatch, strategy)
    for name, img, q, want in make():
        with hotpath.SignatureTable.from_bytes(img) as tab, tab.scan(q, np.array([0, len(q)]), hotpath.Params(aa=True, min_hits=2)) as r:
            got = [(int(h["from0InProt"]), int(h["oI"]), int(h["avgOffFromEnd"]), int(h["fI"]), float(h["functionWt"]))
                   for h in r.hits()]
            assert got == want, (name, strategy, got)
            if strategy != "direct":
                assert r.stats["partitioned"] == 1, name
