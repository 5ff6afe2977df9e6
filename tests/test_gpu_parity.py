"""GPU parity: libkmerguts_hip (through the C ABI) against the CPU oracle, bit-exact.

PARITY STATUS of the oracle itself: unpinned (the reference has no golden vectors for this path
and cannot run here; see oracle/kg_oracle.h).  These tests pin HIP == oracle.
"""
import numpy as np
import pytest
import torch

from helpers import assert_same_records, plant

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["direct", "partitioned", "partitioned_tags"])
def strategy(request, monkeypatch):
    """Every parity test runs with the scan strategies: direct probing; partitioned probing (KG_PARTITION=1 forces the
    bucketed path even on tables small enough for the direct one) -- the byte home index probed in the L2 by scans without
    KG_F_COUNTERS, the tags by scans with them; and the same with the index switched off (KG_BIDX=0: tags for every scan).
    Most workloads here scan with counters (they compare them with the oracle's), which the index path does not serve: under
    "partitioned" every such scan is therefore REPEATED without counters -- the index path -- and its records, event bytes and
    flags must be the ones of the scan the test goes on to compare with the oracle.  Likewise under "direct": without counters
    the direct kernel first asks the table's bit-per-slot digest (scan_kernel, hbits)."""
    monkeypatch.setenv("KG_PARTITION", "0" if request.param == "direct" else "1")
    monkeypatch.setenv("KG_DIRECT_FILTER", "2")          # (by default only tables of more than 4 M slots are asked through the digest)
    if request.param == "partitioned_tags":
        monkeypatch.setenv("KG_BIDX", "0")
    if request.param in ("partitioned", "direct"):
        import dataclasses
        from kmergutsjava_amd import hotpath
        plain = hotpath.SignatureTable.scan
        direct = request.param == "direct"      # the direct kernel without counters asks the table's bit-per-slot digest first

        def scan_and_cross_check(self, seq, offsets, params=None, device_ptr=None):
            r = plain(self, seq, offsets, params, device_ptr)
            if params is not None and params.counters and r.stats["partitioned"] == (0 if direct else 1):
                with plain(self, seq, offsets, dataclasses.replace(params, counters=False), device_ptr) as ri:
                    assert direct or ri.stats["part_levels"] == 4, ri.stats
                    assert ri.stats["partitioned"] == r.stats["partitioned"]
                    for kind in ("hits", "calls", "otu", "hit_events", "container_tail_events", "container_hit_start", "container_call_start"):
                        assert getattr(ri, kind)().tobytes() == getattr(r, kind)().tobytes(), "index path vs tag path: %s differ" % kind
                    for k in ("n_hits", "n_calls", "lookup_ran_off", "fallback"):
                        assert ri.stats[k] == r.stats[k], (k, ri.stats[k], r.stats[k])
            return r
        monkeypatch.setattr(hotpath.SignatureTable, "scan", scan_and_cross_check)
    return request.param


@pytest.fixture(scope="module")
def hp():
    from kmergutsjava_amd import hotpath
    return hotpath


def _img(rec):
    from kmergutsjava_amd import synth
    return synth.table_image(rec)


def test_dna_planted_hits(hp, oracle):
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(50021, 0.6, 7)
    img = _img(rec)
    seq, off = synth.dna_uniform_config(7, 1900, 11)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=37)
    with hp.SignatureTable.from_bytes(img) as tab:
        assert tab.info()["numSigs"] == 50021 and tab.info()["occupied"] == placed
        for mh in (2, 5):
            ora = oracle.run(img, sb, off, lookup_mode=1, min_hits=mh)
            with tab.scan(sb, off, hp.Params(min_hits=mh, counters=True)) as r:
                assert_same_records(r, ora, "dna planted mh=%d" % mh)
                assert r.stats["windows_valid"] == ora["windows_valid"]
                assert r.stats["slots_inspected"] == ora["slots_inspected"]
            assert len(ora["hits"]) > 100


def test_literal_lookup_equals_hip(hp, oracle):
    """HIP direct probing == the reference's sorted merge-join (oracle lookup_mode 0)."""
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(20011, 0.9, 17)      # long probe clusters
    img = _img(rec)
    seq, off = synth.dna_uniform_config(5, 3000, 12)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=29)
    ora = oracle.run(img, sb, off, lookup_mode=0, min_hits=3)
    with hp.SignatureTable.from_bytes(img) as tab, tab.scan(sb, off, hp.Params(min_hits=3)) as r:
        assert_same_records(r, ora, "literal merge-join")


@pytest.mark.parametrize("dna", [True, False])
@pytest.mark.parametrize("oc", [False, True])
def test_high_density_calls_and_otu(hp, oracle, dna, oc):
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(24, 220, 40009, 9000, dna=dna)
    img = _img(rec)
    sb = seq.numpy().tobytes()
    for mh, gap in ((5, 200), (2, 8), (3, 30), (2, 7)):
        ora = oracle.run(img, sb, off, aa=not dna, lookup_mode=1, min_hits=mh, max_gap=gap, order_constraint=oc)
        with hp.SignatureTable.from_bytes(img) as tab:
            with tab.scan(sb, off, hp.Params(aa=not dna, min_hits=mh, max_gap=gap, order_constraint=oc)) as r:
                assert_same_records(r, ora, "high density dna=%s oc=%s mh=%d gap=%d" % (dna, oc, mh, gap))
        if mh == 5 and not oc:
            assert len(ora["calls"]) > 50 and ora["otu"]["n"].max() >= 3


def test_aa_plumbing_slice(hp, oracle):
    from kmergutsjava_amd import synth
    seq, off, rec, placed = synth.plumbing_config(n_seqs=600, num_sigs=100003, n_sigs=50000)
    img = _img(rec)
    sb = seq.numpy().tobytes()
    ora = oracle.run(img, sb, off, aa=True, lookup_mode=1)
    with hp.SignatureTable.from_bytes(img) as tab, tab.scan(sb, off, hp.Params(aa=True, counters=True)) as r:
        assert_same_records(r, ora, "aa plumbing")
        assert r.stats["windows_valid"] == ora["windows_valid"]
        assert r.stats["slots_inspected"] == ora["slots_inspected"]
    assert len(ora["hits"]) > 1000


def test_ragged_short_and_dirty_inputs(hp, oracle):
    """Empty batch, zero-length and sub-window sequences, lowercase / IUPAC / junk characters."""
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(30011, 0.5, 3)
    img = _img(rec)
    rng = np.random.default_rng(5)
    with hp.SignatureTable.from_bytes(img) as tab:
        # empty batch
        for aa in (False, True):
            ora = oracle.run(img, b"", np.zeros(1, np.int64), aa=aa, lookup_mode=1)
            with tab.scan(b"", np.zeros(1, np.int64), hp.Params(aa=aa)) as r:
                assert_same_records(r, ora, "empty batch")
        # DNA: every length 0..80 plus block-boundary lengths, with dirt
        lens = list(range(0, 81)) + [191, 192, 193, 214, 215, 216, 217, 383, 384, 385, 407, 408, 600]
        alphabet = np.frombuffer(b"ACGTacgtuUNnRYxX*-", dtype=np.uint8)
        parts = []
        for L in lens:
            s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
            dirty = rng.random(L) < 0.04
            s[dirty] = rng.choice(alphabet, size=int(dirty.sum()))
            parts.append(s)
        off = synth.offsets_of(np.asarray(lens))
        sb = plant(np.concatenate(parts).tobytes(), off, keys.tolist(), every=31, start=0)
        ora = oracle.run(img, sb, off, lookup_mode=1, min_hits=2)
        with tab.scan(sb, off, hp.Params(min_hits=2, counters=True)) as r:
            assert_same_records(r, ora, "ragged dna")
            assert r.stats["windows_valid"] == ora["windows_valid"]
        # protein: lengths 0..40 + block boundaries, lowercase and X/U/* are invalid residues
        lens = list(range(0, 41)) + [63, 64, 65, 71, 72, 73, 127, 128, 129, 136, 137, 300]
        alphabet = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdxXU*BZ", dtype=np.uint8)
        parts = [rng.choice(alphabet, size=L, p=None) for L in lens]
        off = synth.offsets_of(np.asarray(lens))
        sb = plant(np.concatenate(parts).tobytes(), off, keys.tolist(), every=11, dna=False, start=0)
        ora = oracle.run(img, sb, off, aa=True, lookup_mode=1, min_hits=2)
        with tab.scan(sb, off, hp.Params(aa=True, min_hits=2, counters=True)) as r:
            assert_same_records(r, ora, "ragged aa")
            assert r.stats["windows_valid"] == ora["windows_valid"]
        assert len(ora["hits"]) > 20


def test_table_edges(hp, oracle):
    """No wrap-around at the end of the table, truncated file, file longer than numSigs,
    negative keys (occupied, never matching), fingerprint-sharing keys in one cluster."""
    from kmergutsjava_amd import synth
    import struct
    n = 4099
    rec, placed, keys = synth.random_table(n, 0.97, 23)          # nearly full: clusters reach the end
    img = _img(rec)
    seq, off = synth.dna_uniform_config(4, 2500, 13)
    allk = keys.tolist()
    sb = plant(seq.numpy().tobytes(), off, allk, every=25)
    variants = {
        "full": img,
        "truncated": img[:24 + 24 * 3000 + 7],                       # partial trailing record = EOF
        "longer": img + img[24:24 + 24 * 50],                         # stream continues past numSigs
    }
    # negative key in the middle of a cluster and an early empty slot
    b = bytearray(img)
    b[24 + 24 * 100:24 + 24 * 100 + 8] = struct.pack("<q", -5)
    b[24 + 24 * 2000:24 + 24 * 2000 + 8] = struct.pack("<q", synth.EMPTY_KEY + 12345)
    variants["negative+hole"] = bytes(b)
    for name, im in variants.items():
        ora = oracle.run(im, sb, off, lookup_mode=1, min_hits=2)
        ora0 = oracle.run(im, sb, off, lookup_mode=0, min_hits=2)
        assert ora["hits"].tobytes() == ora0["hits"].tobytes(), name
        with hp.SignatureTable.from_bytes(im) as tab, tab.scan(sb, off, hp.Params(min_hits=2, counters=True)) as r:
            assert_same_records(r, ora, "table " + name)
            assert r.stats["slots_inspected"] == ora["slots_inspected"], name


def test_errors(hp, native):
    import ctypes as C
    with pytest.raises(native.KmerGutsNativeError):
        hp.SignatureTable.from_bytes(b"\x00" * 10)
    bad = np.zeros(48, dtype=np.uint8)
    bad[0] = 1; bad[8] = 16                                       # entrySize 16
    with pytest.raises(native.KmerGutsNativeError):
        hp.SignatureTable.from_bytes(bad)
    from kmergutsjava_amd import synth
    rec, _, _ = synth.random_table(101, 0.5, 1)
    with hp.SignatureTable.from_bytes(synth.table_image(rec)) as tab:
        with pytest.raises(native.KmerGutsNativeError) as ei:
            tab.scan(b"ACGT" * 10, np.array([0, 40]), hp.Params(min_hits=1))
        assert ei.value.code == -6


def test_determinism_and_device_input(hp, oracle):
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(200003, 0.5, 31)
    img = _img(rec)
    seq, off = synth.dna_mix_config(300_000)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=90)
    ora = oracle.run(img, sb, off, lookup_mode=1)
    d = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    with hp.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off) as r1, tab.scan(None, off, device_ptr=d.data_ptr()) as r2:
            assert_same_records(r1, ora, "mix host input")
            assert_same_records(r2, ora, "mix device input")
            assert r1.hits().tobytes() == r2.hits().tobytes()


def _single_function_protein(keys, rec, n_kmers, fn_pick=0, alt_every=0):
    """A protein built from signature 8-mers that all carry one function index (long list, no pair rule)."""
    from kmergutsjava_amd import synth
    r = rec.numpy()
    occupied = r[:, 1] < 5                       # key high word < 5  <=> whichKmer < 5 * 2^32 (occupied)
    fn = r[occupied, 4]
    kk = (r[occupied, 1].astype(np.int64) << 32) | (r[occupied, 0].astype(np.int64) & 0xFFFFFFFF)
    vals, counts = np.unique(fn, return_counts=True)
    f = vals[np.argsort(-counts)[fn_pick]]
    pool = kk[fn == f]
    other = kk[fn != f]
    parts = []
    for i in range(n_kmers):
        if alt_every and i % alt_every == alt_every - 1:
            parts.append(synth.decode_kmer(int(other[i % len(other)])))
        else:
            parts.append(synth.decode_kmer(int(pool[i % len(pool)])))
    return "".join(parts)


@pytest.mark.parametrize("oc", [False, True])
def test_hit_cap_39998(hp, oracle, oc):
    """More than MAX_HITS_PER_SEQ - 2 records in one list (KGJ:496): records beyond the cap are dropped,
    the pair rule keeps being evaluated on the frozen list, and the list later restarts far behind."""
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(2, 50, 20011, 6000, dna=False)
    img = _img(rec)
    prot = (_single_function_protein(keys, rec, 41000) +            # > 39 998 same-function hits, 8 apart
            _single_function_protein(keys, rec, 300, fn_pick=1) +    # a different function right behind
            _single_function_protein(keys, rec, 3000, alt_every=5))
    short = _single_function_protein(keys, rec, 500, fn_pick=2)
    sb = (prot + short).encode()
    off = np.array([0, len(prot), len(prot) + len(short)], dtype=np.int64)
    ora = oracle.run(img, sb, off, aa=True, lookup_mode=1, order_constraint=oc, max_gap=10)
    assert np.diff(ora["container_hit_start"]).max() > 40000
    with hp.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off, hp.Params(aa=True, order_constraint=oc, max_gap=10)) as r:
            assert_same_records(r, ora, "cap oc=%s" % oc)
    if not oc:
        assert ora["calls"]["count"].max() == 39998


@pytest.mark.parametrize("shift", [6, 7, 9])
def test_long_containers_in_pieces(hp, oracle, shift, monkeypatch):
    """gatherHits of a long container is split between waves at records that lie more than maxGap behind their predecessor
    (the list restarts there anyway, KGJ:477-484).  A protein of dense runs -- one of them longer than the 39 998 cap, some
    shorter than minHits, some ending in a pair-rule reset -- separated by stretches without hits, against the oracle, with
    small blocks (many pieces) and the production block size; the same with the pieces switched off; -O never splits."""
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(2, 50, 20011, 6000, dna=False)
    img = _img(rec)
    rng = np.random.default_rng(5)
    runs = []
    for k in range(160):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 40, 77, 300, 1500]))
        runs.append(_single_function_protein(keys, rec, n, fn_pick=int(rng.integers(0, 3)), alt_every=int(rng.choice([0, 0, 3, 5]))))
        runs.append("X" * int(rng.choice([3, 12, 30, 200])))             # 3: no gap (maxGap 10); the others: a gap
    runs.insert(40, _single_function_protein(keys, rec, 41000))          # over the cap, pieces start right behind it
    prot = "".join(runs)
    short = _single_function_protein(keys, rec, 500, fn_pick=2)
    sb = (prot + short).encode()
    off = np.array([0, len(prot), len(prot) + len(short)], dtype=np.int64)
    monkeypatch.setenv("KG_AGG_BLOCK_SHIFT", str(shift))
    for oc in (False, True):
        ora = oracle.run(img, sb, off, aa=True, lookup_mode=1, order_constraint=oc, max_gap=10)
        with hp.SignatureTable.from_bytes(img) as tab:
            with tab.scan(sb, off, hp.Params(aa=True, order_constraint=oc, max_gap=10)) as r:
                assert_same_records(r, ora, "pieces shift=%d oc=%s" % (shift, oc))
                if oc:
                    assert r.stats["agg_pieces"] == 0
                else:
                    assert r.stats["agg_pieces"] >= (30 if shift < 9 else 5), r.stats["agg_pieces"]
                    assert len(ora["calls"]) > 50 and ora["calls"]["count"].max() == 39998
            monkeypatch.setenv("KG_AGG_PIECES", "0")
            with tab.scan(sb, off, hp.Params(aa=True, order_constraint=oc, max_gap=10)) as r:
                assert_same_records(r, ora, "one piece per container oc=%s" % oc)
                assert r.stats["agg_pieces"] == 0
            monkeypatch.delenv("KG_AGG_PIECES")


@pytest.mark.parametrize("shift", [6, 7, 9])
@pytest.mark.parametrize("dna", [True, False])
def test_dense_containers_in_pair_pieces(hp, oracle, shift, dna, monkeypatch):
    """Dense input has no gaps to cut a long container at; it is cut where the pair rule fires with a known outcome instead
    (KGJ:503-508, 441-449; kg_aggregate.hpp piece_starts_kernel): contigs / proteins assembled from signature k-mers in
    same-function runs of 1..12 (thousands of hits per container, a pair-rule reset every dozen), small blocks (many pieces)
    and the production block size, against the oracle -- event bytes included -- and against the same scan without pair
    pieces; several parameter sets (minHits decides which of the fired sets print a CALL)."""
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(5, 2500 if dna else 3000, 20011, 6000, seed=611, dna=dna)
    img = _img(rec)
    sb = seq.numpy().tobytes()
    monkeypatch.setenv("KG_AGG_BLOCK_SHIFT", str(shift))
    with hp.SignatureTable.from_bytes(img) as tab:
        for kw in (dict(), dict(min_hits=2), dict(min_hits=3, min_weighted_hits=2, max_gap=30), dict(min_hits=8, max_gap=1000)):
            ora = oracle.run(img, sb, off, aa=not dna, lookup_mode=1, **kw)
            assert np.diff(ora["container_hit_start"]).max() > 2000 and len(ora["calls"]) > 100
            with tab.scan(sb, off, hp.Params(aa=not dna, **kw)) as r:
                assert_same_records(r, ora, "pair pieces shift=%d dna=%s %s" % (shift, dna, kw))
                pieces = r.stats["agg_pieces"]
                assert pieces >= (40 if shift < 9 else 8), pieces
            monkeypatch.setenv("KG_AGG_PAIRS", "0")
            with tab.scan(sb, off, hp.Params(aa=not dna, **kw)) as r:
                assert_same_records(r, ora, "no pair pieces %s" % kw)
                assert r.stats["agg_pieces"] < pieces
            monkeypatch.delenv("KG_AGG_PAIRS")


def test_otu_votes_in_runs(hp, oracle):
    """The OTU stage replays consecutive voters with the same otuIndex as one step (KGJ:413-439 applied r times = count + r
    and one bubble pass).  Tables whose signatures name very few OTUs, unevenly, so that the voters come in long runs and
    the 5-entry buffer overflows and reorders: OTU records against the oracle."""
    from kmergutsjava_amd import synth
    for n_otu, skew in ((2, 0.9), (3, 0.6), (9, 0.5), (40, 0.3)):
        seq, off, rec, keys = synth.high_density_config(12, 400, 40009, 9000, seed=77 + n_otu, dna=True)
        r_ = rec.numpy()
        rng = np.random.default_rng(n_otu)
        oi = np.where(rng.random(len(r_)) < skew, 0, rng.integers(0, n_otu, len(r_))).astype(np.int32)
        r_[:, 2] = oi                                                      # otuIndex column of the 24-byte records
        img = _img(rec)
        sb = seq.numpy().tobytes()
        for mh in (2, 5):
            ora = oracle.run(img, sb, off, lookup_mode=1, min_hits=mh, max_gap=60)
            with hp.SignatureTable.from_bytes(img) as tab, tab.scan(sb, off, hp.Params(min_hits=mh, max_gap=60)) as r:
                assert_same_records(r, ora, "otu runs n_otu=%d mh=%d" % (n_otu, mh))
            assert len(ora["calls"]) > 20 and ora["otu"]["n"].max() >= min(n_otu, 2)


def test_midsize_persistent_waves(hp, oracle):
    """Enough blocks that every persistent wave walks several of them (grid-stride loop, staging
    chunks handed out across blocks), checked record for record against the oracle."""
    import os
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(3_000_017, 0.5, 41)
    img = _img(rec)
    seq, off = synth.dna_mix_config(12_000_000)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=97)
    ora = oracle.run(img, sb, off, lookup_mode=1)
    assert len(ora["hits"]) > 100000
    with hp.SignatureTable.from_bytes(img) as tab:
        for grid, rpg, chunk in ((64, 3, 16), (256, 6, 512), (2048, 1, 1), (2048, 2, 64)):
            os.environ["KG_SCAN_GRID"] = str(grid); os.environ["KG_SCAN_RPG"] = str(rpg)
            os.environ["KG_STAGE_CHUNK"] = str(chunk)
            try:
                with tab.scan(sb, off, hp.Params(counters=True)) as r:
                    assert_same_records(r, ora, "midsize grid=%d rpg=%d chunk=%d" % (grid, rpg, chunk))
                    assert r.stats["slots_inspected"] == ora["slots_inspected"]
            finally:
                for k in ("KG_SCAN_GRID", "KG_SCAN_RPG", "KG_STAGE_CHUNK"):
                    os.environ.pop(k, None)


def _ecoli(name):
    import gzip, os
    from kmergutsjava_amd.kmer_guts_java import read_fasta
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    ids, seqs = [], []
    read_fasta(gzip.open(path, "rb").read().decode("latin-1"), lambda n, s, d: (ids.append(n), seqs.append(s.encode())))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    return ids, b"".join(seqs), off


def test_ecoli_reference_fixtures(hp, oracle):
    """The reference's own test inputs (test/data/Ecoli_K12_W3110.{faa,fna}.gz; no expected output exists)
    against a synthetic table whose signatures are partly the proteome's own 8-mers, so that the genome's
    genes produce real CALLs in DNA mode.  HIP == oracle (literal merge-join), record for record."""
    from kmergutsjava_amd import synth
    ids_p, prot, off_p = _ecoli("Ecoli_K12_W3110.faa.gz")
    assert len(ids_p) == 13645 and int(off_p[-1]) == 4147102
    ids_g, dna, off_g = _ecoli("Ecoli_K12_W3110.fna.gz")
    assert len(ids_g) == 1 and int(off_g[-1]) == 4646332
    codes = synth.aa_codes(torch.frombuffer(bytearray(prot), dtype=torch.uint8))
    vals = synth.encode_windows_aa(codes)
    vals = vals[vals >= 0]
    own = vals[synth._uniform(71, 0, 400000, int(vals.numel()), "cpu")]
    keys = torch.unique(torch.cat([own, synth.random_keys(400000, 72)]))
    fn = (synth._lsr(synth.splitmix64(73, keys // 20 ** 5), 3) % 500).to(torch.int32)    # neighbours share a function
    otu, avg, _, wt = synth.payload_of(keys, 74, n_otu=12)
    rec, placed = synth.build_table(keys, (otu, avg, fn, wt), 2_000_003)
    img = _img(rec)
    with hp.SignatureTable.from_bytes(img) as tab:
        ora = oracle.run(img, prot, off_p, aa=True, lookup_mode=0)
        with tab.scan(prot, off_p, hp.Params(aa=True, counters=True)) as r:
            assert_same_records(r, ora, "E. coli proteome")
            assert r.stats["windows_valid"] == ora["windows_valid"]
        assert len(ora["hits"]) > 300000
        ora = oracle.run(img, dna, off_g, lookup_mode=0, min_hits=3)
        with tab.scan(dna, off_g, hp.Params(min_hits=3, counters=True)) as r:
            assert_same_records(r, ora, "E. coli genome")
            assert r.stats["windows_valid"] == ora["windows_valid"]
        assert len(ora["calls"]) > 100


def test_cli_gz_data_dir_and_duplicate_ids(tmp_path):
    """KmerGutsJava.main on a gzipped data directory (KGJ:750-758) and a gzipped FASTA (KGJ:764-766);
    a repeated id is reported once, at its first place, with its last record's data (KGJ:772, 805-809)."""
    import gzip
    from oracle import kgj_model as M
    from kmergutsjava_amd import synth, KmerGutsJava
    seq, off, rec, keys = synth.high_density_config(5, 40, 1009, 400, seed=31, dna=True)
    img = synth.table_image(rec)
    sb = seq.numpy().tobytes()
    names = ["a", "b", "a", "c", "b"]
    fa = "".join(">%s x\n%s\n" % (names[k], sb[off[k]:off[k + 1]].decode()) for k in range(5))
    synth.write_data_dir(str(tmp_path / "d"), img, 64, gz=True)
    with gzip.open(tmp_path / "q.fa.gz", "wt") as f:
        f.write(fa)
    KmerGutsJava.main(["-D", str(tmp_path / "d"), "-q", str(tmp_path / "q.fa.gz"), "-o", str(tmp_path / "o.txt"), "-m", "3"])
    fn = ["synthetic function %d" % i for i in range(64)]
    want = M.Model(min_hits=3).run(img, fn, fa)
    got = (tmp_path / "o.txt").read_text()
    assert got == want
    assert got.count("processing ") == 3 and "CALL\t" in got


def test_native_cli_equals_python_mirror_on_ecoli(tmp_path):
    """kmer_guts (C++) and KmerGutsJava.main (Python) over the same data directory and the reference's E. coli
    genome / proteome: identical report bytes, to stdout and to a file, gz and plain inputs, -a and DNA mode."""
    import gzip, os, subprocess, sys, io
    from kmergutsjava_amd import synth, build, KmerGutsJava
    cli = build.build_cli()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ids_p, prot, off_p = _ecoli("Ecoli_K12_W3110.faa.gz")
    codes = synth.aa_codes(torch.frombuffer(bytearray(prot), dtype=torch.uint8))
    vals = synth.encode_windows_aa(codes)
    vals = vals[vals >= 0]
    keys = torch.unique(torch.cat([vals[synth._uniform(81, 0, 300000, int(vals.numel()), "cpu")], synth.random_keys(100000, 82)]))
    fn = (synth._lsr(synth.splitmix64(83, keys // 20 ** 5), 3) % 300).to(torch.int32)
    otu, avg, _, wt = synth.payload_of(keys, 84, n_otu=9)
    rec, _ = synth.build_table(keys, (otu, avg, fn, wt), 1_000_003)
    synth.write_data_dir(str(tmp_path / "d"), synth.table_image(rec), 300, gz=False)
    for name, extra in (("Ecoli_K12_W3110.fna.gz", ["-m", "3"]), ("Ecoli_K12_W3110.faa.gz", ["-a", "-g", "100"])):
        q = os.path.join(gold, name)
        a, b = tmp_path / (name + ".py.txt"), tmp_path / (name + ".cli.txt")
        KmerGutsJava.main(["-D", str(tmp_path / "d"), "-q", q, "-o", str(a)] + extra)
        subprocess.run([cli, "-D", str(tmp_path / "d"), "-q", q, "-o", str(b)] + extra, check=True, stdout=subprocess.DEVNULL)
        ta, tb = a.read_bytes(), b.read_bytes()
        assert ta == tb and ta.count(b"CALL\t") > 10, name
        so = subprocess.run([cli, "-D", str(tmp_path / "d"), "-q", q] + extra, check=True, stdout=subprocess.PIPE).stdout
        assert so == ta
    # flag quirks: -t falls through to "Unknown parameter" and parsing stops (KGJ:605-611, 616-636)
    r = subprocess.run([cli, "-t", "/tmp", "5", "-D", str(tmp_path / "d")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b"Error: Unknown parameter: -t" in r.stdout and b"Usage: kmer_guts" in r.stdout and r.returncode != 0


def test_skewed_repeats_overflow_and_fallback(hp, oracle, monkeypatch, strategy):
    """Homopolymer runs: every window is the same k-mer, so one slot-range bucket receives everything.  The
    partitioned strategy must spill to its overflow list (and, with a tiny list, fall back to direct probing)
    and still return the reference's records; with the k-mer in the table every window is a hit (39 998 cap)."""
    from kmergutsjava_amd import synth
    import torch
    kkk = sum(8 * 20 ** i for i in range(8))                       # "KKKKKKKK" (AAA AAA ...), "FFFFFFFF" on the other strand
    keys = torch.unique(torch.cat([synth.random_keys(20000, 5), torch.tensor([kkk, sum(4 * 20 ** i for i in range(8))])]))
    rec, placed = synth.build_table(keys, synth.payload_of(keys, 6, n_otu=5, n_fn=7), 50021)
    img = _img(rec)
    seq, off = synth.dna_uniform_config(3, 5000, 14)
    parts = [b"A" * 130000, seq.numpy().tobytes()[:5000], b"T" * 40000 + b"ACGT" * 500 + b"A" * 3000, seq.numpy().tobytes()[5000:]]
    sb = b"".join(parts)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.int64)
    ora = oracle.run(img, sb, off, lookup_mode=1)
    assert np.diff(ora["container_hit_start"]).max() > 40000 and ora["calls"]["count"].max() == 39998
    variants = [{}] if strategy == "direct" else [{}, {"KG_PART_MIN_CHUNK_BLOCKS": "1", "KG_PART_CHUNKS": "3"}, {"KG_PART_SLACK": "5"},
                                                  {"KG_PART_SLACK": "5", "KG_PART_OVF_GROUPS": "3"}]
    with hp.SignatureTable.from_bytes(img) as tab:
        for env in variants:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            with tab.scan(sb, off, hp.Params(counters=True)) as r:
                assert_same_records(r, ora, "skew %s %s" % (strategy, env))
                assert r.stats["windows_valid"] == ora["windows_valid"]
                assert r.stats["slots_inspected"] == ora["slots_inspected"]
                if "KG_PART_OVF_GROUPS" in env:      # three overflow groups cannot hold a homopolymer run: thrown away, visibly
                    assert r.stats["partitioned"] == 0 and r.stats["fallback"] == 1, r.stats
                elif strategy == "partitioned":
                    assert r.stats["partitioned"] == 1 and r.stats["fallback"] == 0, r.stats


def test_partitioned_chunk_pipeline(hp, oracle, monkeypatch):
    """The partitioned strategy cuts a batch into chunks of whole sequences and chains their hit ranges on the
    device.  Force up to 8 chunks on small batches: ragged lengths, empty and too-short sequences at the cuts, one
    sequence much longer than a chunk, DNA and protein; -O and a small gap so that CALLs cross block borders."""
    from kmergutsjava_amd import synth
    monkeypatch.setenv("KG_PARTITION", "1")
    monkeypatch.setenv("KG_PART_MIN_CHUNK_BLOCKS", "1")
    keys = synth.random_keys(500_000, 77)
    rec, placed = synth.build_table(keys, synth.payload_of(keys, 76, n_otu=5, n_fn=7), 1_000_003)
    img = _img(rec)
    # DNA: 40 ragged contigs + empties + one long contig in the middle
    lens = [0, 5, 23, 24, 300, 0, 7000, 191, 192, 193, 40000, 0, 0, 815, 100000] + [1000 + 37 * i for i in range(25)] + [0, 23]
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    seq = synth.random_dna(int(off[-1]), 78)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=53)
    n_calls = 0
    for chunks in (2, 3, 5, 8):
        monkeypatch.setenv("KG_PART_CHUNKS", str(chunks))
        for oc, mh, gap in ((False, 2, 200), (True, 2, 30)):
            ora = oracle.run(img, sb, off, order_constraint=oc, min_hits=mh, max_gap=gap, lookup_mode=1)
            n_calls = max(n_calls, len(ora["calls"]))
            with hp.SignatureTable.from_bytes(img) as tab:
                with tab.scan(sb, off, hp.Params(order_constraint=oc, min_hits=mh, max_gap=gap, counters=True)) as r:
                    assert r.stats["partitioned"] == 1
                    assert_same_records(r, ora, "dna chunks=%d oc=%s" % (chunks, oc))
                    assert r.stats["slots_inspected"] == ora["slots_inspected"]
        assert len(ora["hits"]) > 1000 and n_calls > 10
    # protein
    plens = [0, 8, 9, 64, 65, 500, 0, 3000] + [100 + 13 * i for i in range(30)] + [7]
    poff = np.zeros(len(plens) + 1, dtype=np.int64)
    np.cumsum(plens, out=poff[1:])
    pseq = synth.random_protein(int(poff[-1]), 79)
    psb = plant(pseq.numpy().tobytes(), poff, keys.tolist(), every=29, dna=False)
    ora = oracle.run(img, psb, poff, aa=True, min_hits=2, lookup_mode=1)
    assert len(ora["hits"]) > 100
    for chunks in (2, 4, 7):
        monkeypatch.setenv("KG_PART_CHUNKS", str(chunks))
        with hp.SignatureTable.from_bytes(img) as tab:
            with tab.scan(psb, poff, hp.Params(aa=True, min_hits=2)) as r:
                assert r.stats["partitioned"] == 1
                assert_same_records(r, ora, "aa chunks=%d" % chunks)


def test_ordering_knobs_do_not_change_the_records(hp, oracle, monkeypatch):
    """The ordered placement has a staged form (row geometry and output through LDS, groups of up to 1536 records) and a
    direct one, and can run the chunks' orderings on streams of their own: same records either way, including groups that
    are too dense for the staging buffer (every window of a stretch hits)."""
    from kmergutsjava_amd import synth
    monkeypatch.setenv("KG_PARTITION", "1")
    monkeypatch.setenv("KG_PART_MIN_CHUNK_BLOCKS", "1")
    monkeypatch.setenv("KG_PART_CHUNKS", "3")
    keys = synth.random_keys(400_000, 91)
    rec, placed = synth.build_table(keys, synth.payload_of(keys, 92, n_otu=5, n_fn=7), 1_000_003)
    img = _img(rec)
    lens = [120000, 0, 90000, 333, 150000, 70000, 24, 200000]
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    seq = synth.random_dna(int(off[-1]), 93)
    sb = plant(seq.numpy().tobytes(), off, keys.tolist(), every=24)         # dense: 8 hits per row of the +0 frame, 8192 per group of 1024 rows
    ora = oracle.run(img, sb, off, min_hits=2, max_gap=200, lookup_mode=1)
    assert len(ora["hits"]) > 20_000
    for env in ({}, {"KG_PLACE_STAGED": "0"}, {"KG_ORDER_STREAMS": "2"}, {"KG_ORDER_STREAMS": "4", "KG_PLACE_STAGED": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with hp.SignatureTable.from_bytes(img) as tab:
            with tab.scan(sb, off, hp.Params(min_hits=2, max_gap=200)) as r:
                assert r.stats["partitioned"] == 1
                assert_same_records(r, ora, "ordering knobs %r" % (env,))
        for k in env:
            monkeypatch.delenv(k)


def test_list_resize_and_rerun(hp, oracle, monkeypatch, strategy):
    """The hit / candidate lists (partitioned) and the staging area (direct) are sized from the hit rate seen so far;
    when they are too small the scan is re-run once with the exact size.  Force that path."""
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(12, 150, 200003, 5000, seed=91, dna=True)
    img = _img(rec)
    sb = seq.numpy().tobytes()
    ora = oracle.run(img, sb, off, min_hits=3, lookup_mode=1)
    assert len(ora["hits"]) > 1000 and len(ora["calls"]) > 50
    monkeypatch.setenv("KG_PART_MIN_CHUNK_BLOCKS", "1")
    monkeypatch.setenv("KG_PART_CHUNKS", "3")
    with hp.SignatureTable.from_bytes(img) as tab:
        monkeypatch.setenv("KG_TEST_TINY_LISTS", "1")
        with tab.scan(sb, off, hp.Params(min_hits=3, counters=True)) as r:
            assert r.stats["scan_launches"] == 2, r.stats["scan_launches"]
            assert_same_records(r, ora, "resized " + strategy)
            assert r.stats["slots_inspected"] == ora["slots_inspected"] and r.stats["windows_valid"] == ora["windows_valid"]
        monkeypatch.delenv("KG_TEST_TINY_LISTS")
        with tab.scan(sb, off, hp.Params(min_hits=3)) as r:              # the table remembers the hit rate: one launch
            assert r.stats["scan_launches"] == 1
            assert_same_records(r, ora, "after resize " + strategy)


def test_low_complexity_protein(hp, oracle, monkeypatch, strategy):
    """Protein input with long single-residue and two-residue repeats (the scatter pass sets such blocks aside for
    lowc_blocks_kernel); the repeated k-mers are signatures, so the hit lists run into the 39 998 cap."""
    from kmergutsjava_amd import synth
    import torch
    kk = sum(8 * 20 ** i for i in range(8))                            # KKKKKKKK
    kr = sum((8 if i % 2 else 14) * 20 ** i for i in range(8))         # KRKRKRKR / RKRKRKRK differ: add both
    rk = sum((14 if i % 2 else 8) * 20 ** i for i in range(8))
    keys = torch.unique(torch.cat([synth.random_keys(30000, 15), torch.tensor([kk, kr, rk])]))
    rec, placed = synth.build_table(keys, synth.payload_of(keys, 16, n_otu=5, n_fn=7), 100003)
    img = _img(rec)
    rnd = synth.random_protein(6000, 17).numpy().tobytes()
    parts = [b"K" * 50000, rnd[:3000], b"KR" * 21000 + b"MKV" + b"K" * 900, rnd[3000:], b"", b"R" * 70]
    sb = b"".join(parts)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.int64)
    ora = oracle.run(img, sb, off, aa=True, min_hits=2, lookup_mode=1)
    assert np.diff(ora["container_hit_start"]).max() > 40000 and ora["calls"]["count"].max() == 39998
    for env in ({}, {"KG_PART_MIN_CHUNK_BLOCKS": "1", "KG_PART_CHUNKS": "2"}, {"KG_PART_SLACK": "5"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with hp.SignatureTable.from_bytes(img) as tab:
            with tab.scan(sb, off, hp.Params(aa=True, min_hits=2, counters=True)) as r:
                assert_same_records(r, ora, "low-complexity protein %s %s" % (strategy, env))
                assert r.stats["windows_valid"] == ora["windows_valid"]
                assert r.stats["slots_inspected"] == ora["slots_inspected"]
