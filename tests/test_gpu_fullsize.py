"""BASELINE-size configurations through the C ABI on the MI355X: the production geometry of the partitioned scan
(33.6 GB table, 2^21-slot buckets, 668 of them, chunks of whole contigs on three streams) is reached only at these
sizes.  The oracle cannot finish 1 Gbp in seconds, so parity is proven in three ways:

  * the two scan strategies (direct probing / partitioned probing: different kernels, different orderings of the
    work) must leave byte-identical hit, CALL and OTU records and container offsets in HBM -- compared on the device;
  * the oracle (literal sorted merge-join, lookup_mode 0) scans a sub-batch of whole contigs drawn from EVERY chunk of
    the pipeline (sequences are independent, KGJ:528, 540) and those containers' records must be byte-identical;
  * size-independent properties of the whole result: hits strictly ascending in (container, from0InProt),
    container_hit_start consistent with the records, the oracle's own counters reproduced on the sample.

PARITY STATUS of the oracle itself: unpinned (oracle/kg_oracle.h) -- the reference holds no input -> output pair.
"""
import struct

import numpy as np
import pytest
import torch

from helpers import chunk_seq_ranges

pytestmark = pytest.mark.gpu

NUM_SIGS = 1_400_303_159          # BASELINE.md section 4 / SURVEY 8d: the full-size table, 33.6 GB at load 0.5


@pytest.fixture(scope="module")
def hp():
    from kmergutsjava_amd import hotpath
    return hotpath


@pytest.fixture(scope="module")
def full_table(hp):
    """The 1 400 303 159-slot table, built in HBM (seed 202, as bench.py), plus a lazily made host image for the oracle."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    rec, placed, keys = synth.random_table(NUM_SIGS, 0.5, 202, dev)
    del keys
    torch.cuda.synchronize()
    tab = hp.SignatureTable.from_device_ptr(rec.data_ptr(), NUM_SIGS, 0, keepalive=rec)
    assert tab.info()["occupied"] == placed
    box = {"tab": tab, "rec": rec, "img": None}

    def image():
        if box["img"] is None:
            host = torch.empty(24 + NUM_SIGS * 24, dtype=torch.uint8)
            host[:24] = torch.frombuffer(bytearray(struct.pack("<qqq", NUM_SIGS, 24, 1)), dtype=torch.uint8)
            host[24:].view(torch.int32).view(NUM_SIGS, 6).copy_(rec)
            box["img"] = host.numpy()
        return box["img"]

    box["image"] = image
    yield box
    tab.close()
    box["img"] = None
    del rec
    torch.cuda.empty_cache()


def _same_on_device(a, b, what):
    for name in ("hits", "calls", "otu", "container_hit_start", "container_call_start"):
        x, y = a.device_view(name), b.device_view(name)
        assert x.shape == y.shape, "%s: %s sizes differ (%s vs %s)" % (what, name, tuple(x.shape), tuple(y.shape))
        assert torch.equal(x, y), "%s: %s differ between the strategies" % (what, name)
    for k in ("n_hits", "n_calls", "residues", "windows"):
        assert a.stats[k] == b.stats[k], (what, k)


def _whole_result_properties(r, per):
    """hits[] ordered by (container, from0InProt), positions unique per container, offsets consistent."""
    n = r.stats["n_hits"]
    h = r.device_view("hits").view(torch.int32).view(-1, 6)
    cont = h[:, 0].to(torch.int64) & 0xFFFFFFFF
    key = (cont << 32) | (h[:, 1].to(torch.int64) & 0xFFFFFFFF)
    assert int(h[:, 1].min()) >= 0
    assert bool((key[1:] > key[:-1]).all()), "hits not strictly ascending in (container, from0InProt)"
    chs = r.device_view("container_hit_start")
    n_cont = r.stats["n_containers"]
    assert int(chs[0]) == 0 and int(chs[-1]) == n
    want = torch.searchsorted(cont.contiguous(), torch.arange(n_cont + 1, dtype=torch.int64, device=cont.device))
    assert torch.equal(chs, want), "container_hit_start does not match the records"
    ccs = r.device_view("container_call_start")
    assert int(ccs[0]) == 0 and int(ccs[-1]) == r.stats["n_calls"] and bool((ccs[1:] >= ccs[:-1]).all())
    assert n_cont == r.stats["n_seqs"] * per


def _oracle_sample(oracle, image, seq, off, idx, r, what, **kw):
    """The oracle on the sub-batch idx (literal merge-join) against the same containers of the full-size result."""
    sub_off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum((off[1:] - off[:-1])[idx], out=sub_off[1:])
    sub = torch.cat([seq[int(off[i]):int(off[i + 1])] for i in idx]).cpu().numpy()
    ora = oracle.run(image, sub, sub_off, lookup_mode=0, **kw)
    got = r.subset(idx, events=True)
    for name in ("hits", "calls", "otu", "hit_events", "container_tail_events"):
        assert got[name].tobytes() == ora[name].tobytes(), "%s: %s of the sampled contigs differ from the oracle" % (what, name)
    assert np.array_equal(got["container_hit_start"], ora["container_hit_start"]), what
    assert np.array_equal(got["container_call_start"], ora["container_call_start"]), what
    return ora


def test_config3_1gbp_contig_mix_full_table(hp, oracle, full_table, monkeypatch):
    """BASELINE config 3 exactly as bench.py runs it: 1 Gbp contig mix vs the 33.6 GB table, default environment."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    lens = synth.contig_mix_lengths(1_000_000_000, 301)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 302, dev)
    torch.cuda.synchronize()
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as rp:
        st = rp.stats
        assert st["partitioned"] == 1 and st["fallback"] == 0 and st["scan_launches"] >= 1
        assert st["part_chunks"] == 4 and st["part_shift"] == 21 and st["part_buckets"] == 668, st
        _whole_result_properties(rp, 6)
        # contigs from every one of the four chunks
        idx = synth.spread_sample(off, groups=4, per_group=20, max_bp_per_group=2_500_000)
        ranges = chunk_seq_ranges(off, 4)
        per_chunk = [int(((idx >= a) & (idx < b)).sum()) for a, b in ranges]
        assert len(ranges) == 4 and min(per_chunk) >= 3, (ranges, per_chunk)
        ora = _oracle_sample(oracle, full_table["image"](), seq, off, idx, rp, "config 3 partitioned")
        assert len(ora["hits"]) > 100_000
        # the instrumented kernels (KG_F_COUNTERS) and the other strategy: identical records on the device
        with tab.scan(None, off, hp.Params(counters=True), device_ptr=seq.data_ptr()) as rc:
            assert rc.stats["partitioned"] == 1 and rc.stats["fallback"] == 0
            _same_on_device(rp, rc, "config 3 partitioned vs partitioned+counters")
            wv, si = rc.stats["windows_valid"], rc.stats["slots_inspected"]
        monkeypatch.setenv("KG_PARTITION", "0")
        with tab.scan(None, off, hp.Params(counters=True), device_ptr=seq.data_ptr()) as rd:
            assert rd.stats["partitioned"] == 0 and rd.stats["fallback"] == 0
            _same_on_device(rp, rd, "config 3 partitioned vs direct")
            assert rd.stats["windows_valid"] == wv and rd.stats["slots_inspected"] == si
        # the same counters from the oracle on the sample (direct-probe mode counts slots; merge-join mode does not)
        sub_off = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum((off[1:] - off[:-1])[idx], out=sub_off[1:])
        sub = torch.cat([seq[int(off[i]):int(off[i + 1])] for i in idx])
        o1 = oracle.run(full_table["image"](), sub.cpu().numpy(), sub_off, lookup_mode=1)
        monkeypatch.delenv("KG_PARTITION", raising=False)
        with tab.scan(None, sub_off, hp.Params(counters=True), device_ptr=sub.data_ptr()) as rs:
            assert rs.stats["windows_valid"] == o1["windows_valid"] and rs.stats["slots_inspected"] == o1["slots_inspected"]
            assert rs.hits().tobytes() == o1["hits"].tobytes() == ora["hits"].tobytes()


def test_aggregation_in_pieces_equals_one_wave_per_container_at_full_size(hp, full_table, monkeypatch):
    """Long containers are aggregated in pieces that start behind gaps > maxGap (kg_aggregate.hpp).  At BASELINE size --
    1 Gbp of contigs up to 1 Mbp -- and for ONE contig of 120 Mbp (six containers of 700 k hits) the CALL, OTU and event
    records must be the ones of the one-wave-per-container walk (KG_AGG_PIECES=0), byte for byte."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    lens = synth.contig_mix_lengths(1_000_000_000, 301)
    for what, off in (("1 Gbp contig mix", synth.offsets_of(lens)), ("one contig of 120 Mbp", np.array([0, 120_000_000], dtype=np.int64))):
        seq = synth.random_dna(int(off[-1]), 302, dev)
        torch.cuda.synchronize()
        for mh, gap in ((5, 200), (2, 40)):
            monkeypatch.delenv("KG_AGG_PIECES", raising=False)
            with tab.scan(None, off, hp.Params(min_hits=mh, max_gap=gap), device_ptr=seq.data_ptr()) as a:
                assert a.stats["agg_pieces"] > 1000, (what, a.stats["agg_pieces"])
                monkeypatch.setenv("KG_AGG_PIECES", "0")
                with tab.scan(None, off, hp.Params(min_hits=mh, max_gap=gap), device_ptr=seq.data_ptr()) as b:
                    assert b.stats["agg_pieces"] == 0
                    _same_on_device(a, b, what + " pieces vs none")
                    assert np.array_equal(a.hit_events(), b.hit_events()), what + ": hit events"
                    assert np.array_equal(a.container_tail_events(), b.container_tail_events()), what + ": tail events"
                    if mh == 2:
                        assert a.stats["n_calls"] > 1000, (what, a.stats["n_calls"])
        del seq
        torch.cuda.empty_cache()


def test_config2_100mbp_uniform_full_table(hp, oracle, full_table, monkeypatch):
    """BASELINE config 2: 1000 x 100 kbp uniform DNA vs the full table (partitioned by default, one or two chunks)."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    seq, off = synth.dna_uniform_config(1000, 100_000, 201, dev)
    torch.cuda.synchronize()
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as rp:
        assert rp.stats["partitioned"] == 1 and rp.stats["fallback"] == 0, rp.stats
        _whole_result_properties(rp, 6)
        idx = synth.spread_sample(off, groups=max(4, rp.stats["part_chunks"]), per_group=5)
        _oracle_sample(oracle, full_table["image"](), seq, off, idx, rp, "config 2")
        monkeypatch.setenv("KG_PARTITION", "0")
        with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as rd:
            assert rd.stats["partitioned"] == 0
            _same_on_device(rp, rd, "config 2 partitioned vs direct")
    # the other parameters of the CLI (-O, -m, -M, -g) at this size: oracle on the same sample
    monkeypatch.delenv("KG_PARTITION", raising=False)
    for kw in (dict(order_constraint=True, min_hits=2, max_gap=600), dict(min_hits=3, min_weighted_hits=2, max_gap=50),
               dict(min_hits=2, max_gap=2_000_000_000)):
        with tab.scan(None, off, hp.Params(**kw), device_ptr=seq.data_ptr()) as r:
            assert r.stats["fallback"] == 0 and (r.stats["agg_pieces"] == 0 or not kw.get("order_constraint"))
            _oracle_sample(oracle, full_table["image"](), seq, off, idx, r, "config 2 %s" % kw, **kw)


@pytest.mark.parametrize("dna", [True, False])
def test_config5_high_density_full_size(hp, oracle, dna, monkeypatch):
    """BASELINE config 5: 100 Mbp (DNA) / 10 k proteins assembled from signature k-mers of <= 32 functions, <= 8 OTUs."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    n_contigs, kpc = (1000, 4167) if dna else (10000, 38)
    seq, off, rec = synth.high_density_device(n_contigs, kpc, 20_000_003, 8_000_000, 501, dna, dev)
    torch.cuda.synchronize()
    img = synth.table_image(rec)
    per = 6 if dna else 1
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with hp.SignatureTable.from_device_ptr(rec.data_ptr(), 20_000_003, 0, keepalive=rec) as tab:
        with tab.scan(None, off, hp.Params(aa=not dna), device_ptr=seq.data_ptr()) as ra:
            assert ra.stats["fallback"] == 0
            _whole_result_properties(ra, per)
            assert ra.stats["n_calls"] > (100_000 if dna else 5_000) and ra.stats["n_hits"] > 10 * ra.stats["n_calls"]
            idx = synth.spread_sample(off, groups=4, per_group=10 if dna else 500, max_bp_per_group=1_000_000)
            ora = _oracle_sample(oracle, img, seq, off, idx, ra, "config 5 dna=%s" % dna, aa=not dna)
            assert len(ora["calls"]) > 100
            monkeypatch.setenv("KG_PARTITION", "0" if ra.stats["partitioned"] else "1")
            with tab.scan(None, off, hp.Params(aa=not dna), device_ptr=seq.data_ptr()) as rb:
                assert rb.stats["partitioned"] != ra.stats["partitioned"] and rb.stats["fallback"] == 0, rb.stats
                _same_on_device(ra, rb, "config 5 dna=%s, the two strategies" % dna)
            # ra ran the direct kernel behind the table's bit-per-slot digest (20 M slots: the default there); the same without
            monkeypatch.setenv("KG_PARTITION", "0")
            monkeypatch.setenv("KG_DIRECT_FILTER", "0")
            with tab.scan(None, off, hp.Params(aa=not dna), device_ptr=seq.data_ptr()) as rc:
                assert rc.stats["partitioned"] == 0
                _same_on_device(ra, rc, "config 5 dna=%s, direct with and without the digest" % dna)
                assert rc.stats["lookup_ran_off"] == ra.stats["lookup_ran_off"]


def test_table_with_more_than_2_31_slots(hp, oracle):
    """numSigs >= 2^31: slots and k-mer quotients no longer fit the 32-bit arithmetic of the partitioned strategy
    (kg::split_fast), so the scan runs the direct kernel with 64-bit slot arithmetic (kg::split_value) -- against a 51.5 GB
    table, records byte-identical to the oracle's (direct-probe mode), counters included."""
    from kmergutsjava_amd import synth
    from helpers import assert_same_records
    dev = torch.device("cuda", 0)
    n = (1 << 31) + 11
    rec, placed, keys = synth.random_table(n, 0.3, 909, dev)
    del keys
    torch.cuda.synchronize()
    lens = synth.contig_mix_lengths(20_000_000, 77)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 78, dev)
    torch.cuda.synchronize()
    host = torch.empty(24 + n * 24, dtype=torch.uint8)
    host[:24] = torch.frombuffer(bytearray(struct.pack("<qqq", n, 24, 1)), dtype=torch.uint8)
    host[24:].view(torch.int32).view(n, 6).copy_(rec)
    ora = oracle.run(host.numpy(), seq.cpu().numpy(), off, lookup_mode=1)
    assert len(ora["hits"]) > 100_000 and int(ora["hits"]["container"].max()) > 100
    with hp.SignatureTable.from_device_ptr(rec.data_ptr(), n, 0, keepalive=rec) as tab:
        assert tab.info()["numSigs"] == n and tab.info()["occupied"] == placed
        with tab.scan(None, off, hp.Params(counters=True), device_ptr=seq.data_ptr()) as r:
            assert r.stats["partitioned"] == 0
            assert_same_records(r, ora, "2^31 + 11 slots")
            assert r.stats["windows_valid"] == ora["windows_valid"] and r.stats["slots_inspected"] == ora["slots_inspected"]
    del host, rec
    torch.cuda.empty_cache()


def test_largest_batch_and_the_limit_behind_it(hp, oracle, full_table):
    """A 2 Gbp batch (3.99e9 of the 2^32 - 256 windows one call takes; more than the 2^23 window blocks of the partitioned
    strategy, so the direct kernel runs): properties of the whole result and the oracle on contigs from its start, middle
    and end.  One more contig pushes the batch over the limit: KG_ERR_LIMIT before anything is touched."""
    from kmergutsjava_amd import synth, _native
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    lens = synth.contig_mix_lengths(2_000_000_000, 401)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 402, dev)
    torch.cuda.synchronize()
    with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as r:
        st = r.stats
        assert st["partitioned"] == 0 and st["fallback"] == 0 and st["windows"] > 3_900_000_000 and st["n_hits"] > 70_000_000, st
        _whole_result_properties(r, 6)
        idx = synth.spread_sample(off, groups=3, per_group=12, max_bp_per_group=1_500_000)
        assert idx[0] < len(lens) // 3 and idx[-1] > len(lens) * 2 // 3
        _oracle_sample(oracle, full_table["image"](), seq, off, idx, r, "2 Gbp batch")
    big = np.concatenate([lens, [300_000_000]])          # 2.3 Gbp: 4.6e9 windows
    off_big = synth.offsets_of(big)
    with pytest.raises(_native.KmerGutsNativeError) as ei:
        tab.scan(None, off_big, hp.Params(), device_ptr=seq.data_ptr())     # (refused from the offsets alone)
    assert ei.value.code == -7
    del seq
    torch.cuda.empty_cache()


def test_largest_partitioned_batch_and_protein_at_scale(hp, oracle, full_table, monkeypatch):
    """The partitioned strategy at its upper end -- 1.55 Gbp, 8.07 M of the 2^23 window blocks it takes, four chunks of
    390 Mbp -- and in protein mode at scale (300 M residues in 330-residue proteins: one window row per block, no strands):
    both against the direct kernel on the device, plus the oracle on proteins drawn from every chunk."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    # -- DNA, 1.55 Gbp
    lens = synth.contig_mix_lengths(1_550_000_000, 501)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 502, dev)
    torch.cuda.synchronize()
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as rp:
        assert rp.stats["partitioned"] == 1 and rp.stats["fallback"] == 0 and rp.stats["part_chunks"] == 4, rp.stats
        assert 8_000_000 < rp.stats["n_blocks"] <= (1 << 23)
        _whole_result_properties(rp, 6)
        monkeypatch.setenv("KG_PARTITION", "0")
        with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as rd:
            assert rd.stats["partitioned"] == 0
            _same_on_device(rp, rd, "1.55 Gbp partitioned vs direct")
    del seq
    torch.cuda.empty_cache()
    # -- protein, 300 M residues
    lens = np.full(300_000_000 // 330, 330, dtype=np.int64)
    lens[::9] = 1200
    lens[::31] = 7                                        # too short for a window
    off = synth.offsets_of(lens)
    seq = synth.random_protein(int(off[-1]), 503, dev)
    torch.cuda.synchronize()
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with tab.scan(None, off, hp.Params(aa=True, min_hits=2), device_ptr=seq.data_ptr()) as rp:
        assert rp.stats["partitioned"] == 1 and rp.stats["fallback"] == 0, rp.stats
        _whole_result_properties(rp, 1)
        idx = synth.spread_sample(off, groups=max(1, rp.stats["part_chunks"]), per_group=400, max_bp_per_group=400_000)
        ora = _oracle_sample(oracle, full_table["image"](), seq, off, idx, rp, "protein at scale", aa=True, min_hits=2)
        assert len(ora["hits"]) > 5_000
        monkeypatch.setenv("KG_PARTITION", "0")
        with tab.scan(None, off, hp.Params(aa=True, min_hits=2), device_ptr=seq.data_ptr()) as rd:
            assert rd.stats["partitioned"] == 0
            _same_on_device(rp, rd, "300 M residues partitioned vs direct")
    del seq
    torch.cuda.empty_cache()


def test_sharded_scans_restored_on_the_device_equal_the_unsharded_scan(hp):
    """The exchange step of the multi-GPU layer without a process group: the batch is cut into three shards of whole
    contigs (distributed.shard_sequences), every shard is scanned on this GPU, the library's own HBM buffers are wrapped
    as torch tensors (ScanResult.device_view: what travels over RCCL) and distributed.restore_hits puts the records in
    global order on the device -- byte-identical to the hit array of the unsharded scan."""
    from kmergutsjava_amd import synth, distributed as kd
    dev = torch.device("cuda", 0)
    rec, placed, keys = synth.random_table(30_000_019, 0.5, 77, dev)
    del keys
    lens = synth.contig_mix_lengths(40_000_000, 301)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 302, dev)
    torch.cuda.synchronize()
    with hp.SignatureTable.from_device_ptr(rec.data_ptr(), 30_000_019, 0, keepalive=rec) as tab:
        with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as whole:
            want_hits = whole.device_view("hits").clone()
            want_chs = whole.device_view("container_hit_start").clone()
        shards = kd.shard_sequences(lens, 3)
        results, hits, chs, idx = [], [], [], []
        for mine in shards:
            s_lens = lens[mine]
            s_off = synth.offsets_of(s_lens)
            s_seq = synth.random_dna_at(off[mine], s_lens, 302, dev)          # the shard's contigs have the batch's bases
            torch.cuda.synchronize()
            r = tab.scan(None, s_off, hp.Params(), device_ptr=s_seq.data_ptr())
            results.append(r)
            hits.append(r.device_view("hits"))
            chs.append(r.device_view("container_hit_start"))
            idx.append(torch.as_tensor(mine, device=dev))
            assert hits[-1].is_cuda and hits[-1].data_ptr() == r.device_hits_ptr()       # zero-copy
        got_hits, got_chs = kd.restore_hits(hits, chs, idx, len(lens), 6)
        assert got_hits.is_cuda
        assert torch.equal(got_hits.contiguous().view(torch.uint8).reshape(-1), want_hits)
        assert torch.equal(got_chs, want_chs)
        assert int(want_chs[-1]) > 10_000
        hits, chs = [], []
        tab.close()                         # results still open: the table closes them first
        assert all(r._h.value is None for r in results)


def test_copy_hits_and_zero_copy_views(hp):
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(40, 300, 200_003, 60_000, dna=True)
    with hp.SignatureTable.from_bytes(synth.table_image(rec)) as tab, tab.scan(seq.numpy(), off, hp.Params(min_hits=2)) as r:
        h = r.hits()
        assert len(h) > 5_000
        assert r.copy_hits().tobytes() == h.tobytes()
        part = np.zeros(1000, dtype=h.dtype)
        assert r.copy_hits(123, 1000, out=part).tobytes() == h[123:1123].tobytes()
        assert r.calls(copy=False).tobytes() == r.calls().tobytes() and r.otu(copy=False).tobytes() == r.otu().tobytes()
        with pytest.raises(Exception):
            r.copy_hits(len(h) - 10, 11)


@pytest.mark.parametrize("strategy", ["direct", "partitioned", "partitioned_tags"])
def test_config1_plumbing_at_its_stated_size(hp, oracle, strategy, monkeypatch):
    """BASELINE config 1 as written: 10 000 proteins of ~300 aa against a 1 000 003-slot table holding 500 000 signatures
    (half of them the sequences' own 8-mers), AA mode, through the whole C ABI: every record against the oracle's literal
    merge-join, with every scan strategy."""
    from kmergutsjava_amd import synth
    from helpers import assert_same_records
    seq, off, rec, placed = synth.plumbing_config()
    assert len(off) - 1 == 10000 and rec.shape[0] == 1000003 and 480000 < placed <= 500000
    img = synth.table_image(rec)
    sb = seq.numpy().tobytes()
    monkeypatch.setenv("KG_PARTITION", "0" if strategy == "direct" else "1")
    monkeypatch.setenv("KG_BIDX", "0" if strategy == "partitioned_tags" else "1")
    monkeypatch.setenv("KG_DIRECT_FILTER", "2")      # direct, no counters: behind the bit-per-slot digest (default: tables > 4 M slots only)
    # the same keys with ONE function per protein for the signatures drawn from it: the reference's defaults then CALL
    seq_c, off_c, rec_c, placed_c = synth.plumbing_config(coherent=True)
    assert placed_c == placed and np.array_equal(off_c, off)
    img_c = synth.table_image(rec_c)
    for image, settings, min_calls in ((img, (dict(), dict(min_hits=2, max_gap=600)), 200), (img_c, (dict(),), 1000)):
      with hp.SignatureTable.from_bytes(image) as tab:
        # the reference's defaults (-m 5 -g 200: with functions hashed from the k-mer hardly any CALL -- that leg checks the
        # hit records; with one function per protein thousands) and a setting that calls a lot on the hashed functions
        for kw in settings:
            ora = oracle.run(image, sb, off, aa=True, lookup_mode=0, **kw)
            assert ora["residues"] > 2_900_000 and len(ora["hits"]) > 100_000
            assert (not kw and image is img) or len(ora["calls"]) > min_calls, len(ora["calls"])
            for counters in (True, False):           # (with the second level: the tag kernels / the home-index kernel)
                with tab.scan(sb, off, hp.Params(aa=True, counters=counters, **kw)) as r:
                    assert_same_records(r, ora, "config 1 %s counters=%s %s" % (strategy, counters, kw))
                    if counters:
                        o1 = oracle.run(image, sb, off, aa=True, lookup_mode=1, **kw)
                        assert r.stats["windows_valid"] == o1["windows_valid"] and r.stats["slots_inspected"] == o1["slots_inspected"]


def test_byte_home_index_against_the_tags_at_full_size(hp, oracle, full_table, monkeypatch):
    """BASELINE config 3 through the two things the tag pass can probe: the table's byte home index (the default; exact for
    this table: quotients 0..18) and the tags (KG_BIDX=0, and every KG_F_COUNTERS scan): the same records over the whole
    1 Gbp result, compared on the device, the oracle's on contigs from every chunk, and the same counters either way."""
    from kmergutsjava_amd import synth
    dev = torch.device("cuda", 0)
    tab = full_table["tab"]
    lens = synth.contig_mix_lengths(1_000_000_000, 301)
    off = synth.offsets_of(lens)
    seq = synth.random_dna(int(off[-1]), 302, dev)
    torch.cuda.synchronize()
    monkeypatch.delenv("KG_PARTITION", raising=False)
    with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as r1:
        assert r1.stats["partitioned"] == 1 and r1.stats["part_levels"] == 4 and r1.stats["fallback"] == 0, r1.stats
        idx = synth.spread_sample(off, groups=4, per_group=10, max_bp_per_group=1_500_000)
        _oracle_sample(oracle, full_table["image"](), seq, off, idx, r1, "byte home index")
        monkeypatch.setenv("KG_BIDX", "0")
        with tab.scan(None, off, hp.Params(), device_ptr=seq.data_ptr()) as r0:
            assert r0.stats["part_levels"] == 1 and r0.stats["fallback"] == 0, r0.stats
            _same_on_device(r1, r0, "byte home index vs tags")
            assert r0.stats["lookup_ran_off"] == r1.stats["lookup_ran_off"]
        with tab.scan(None, off, hp.Params(counters=True), device_ptr=seq.data_ptr()) as rc0:
            c0 = (rc0.stats["windows_valid"], rc0.stats["slots_inspected"])
        monkeypatch.delenv("KG_BIDX")
        with tab.scan(None, off, hp.Params(counters=True), device_ptr=seq.data_ptr()) as rc:
            assert rc.stats["part_levels"] == 1                       # counters: the tag kernels either way
            _same_on_device(r1, rc, "byte home index vs tags with counters")
            assert (rc.stats["windows_valid"], rc.stats["slots_inspected"]) == c0


@pytest.mark.parametrize("n", [900_000_011, 1_400_000_029])
def test_home_index_on_hand_made_clusters_in_a_full_size_table(hp, oracle, monkeypatch, n):
    """The byte home index on the cases it could get wrong, planted by hand into an otherwise empty full-size table and
    queried as proteins: two keys sharing one home slot (a pair code), three and five (hashed codes: a candidate the walk has
    to refute, a certain miss), a key behind a NEGATIVE whichKmer, a key behind a hole (not reachable), a key in front of its
    home slot (not reachable), the same key twice in one run (the first one wins), and keys in the occupied run that ends at the
    end of the record stream (lookup_ran_off).  Oracle: literal merge-join and direct probing.
    n = 900 000 011: quotients up to 28, folded into the index's 19 classes (every listed class is only a candidate);
    n = 1 400 000 029: quotients up to 18 -- a class is the quotient, codes 1..190 are exact (the KmerGuts table's regime)."""
    from kmergutsjava_amd import synth
    import kat_cases as K
    dev = torch.device("cuda", 0)
    rec = torch.empty((n, 6), dtype=torch.int32, device=dev)
    empty = 20 ** 8 + 1
    rec[:, 0] = empty & 0xFFFFFFFF if (empty & 0xFFFFFFFF) < 2 ** 31 else (empty & 0xFFFFFFFF) - 2 ** 32
    rec[:, 1] = empty >> 32
    rec[:, 2:] = 0

    def put(slot, key, oi, fi, wt=1.0):
        lo, hi = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
        row = torch.tensor([lo if lo < 2 ** 31 else lo - 2 ** 32, hi if hi < 2 ** 31 else hi - 2 ** 32, oi, 3, fi,
                            int(np.float32(wt).view(np.int32))], dtype=torch.int32, device=dev)
        rec[slot] = row

    queries = []                                   # (k-mer value, expected oI or None)
    h = 1000                                       # five keys homed at slot 1000: quotients 0..4 (a hashed code: class bits 0..4)
    for q in range(5):
        put(h + q, q * n + h, 10 + q, 2)
        queries.append((q * n + h, 10 + q))
    queries.append((7 * n + h, None))              # same home, unknown quotient (7 % 6 = 1: its bit is set): a candidate, walked, not found
    h = 5000                                       # negative key at the home slot, the real key behind it
    put(h, -12345, 1, 1); put(h + 1, 3 * n + h, 21, 2)
    queries.append((3 * n + h, 21))
    h = 9000                                       # key behind a hole: slot h occupied by another home's key, h+1 empty, h+2 the key
    put(h, 2 * n + h - 1 + 1, 1, 1)                # (a key homed at h with quotient 2)
    put(h + 2, 4 * n + h, 22, 2)
    queries.append((4 * n + h, None)); queries.append((2 * n + h, 1))
    h = 13000                                      # key stored in front of its home slot
    put(h - 1, 5 * n + h, 23, 2)
    queries.append((5 * n + h, None))
    h = 17000                                      # the same key twice in one run: the first record's payload
    put(h, 6 * n + h, 24, 2); put(h + 1, 6 * n + h, 25, 2)
    queries.append((6 * n + h, 24))
    h = 21000                                      # two keys at one home slot (the byte index's pair codes), a third quotient absent
    put(h, 3 * n + h, 31, 2); put(h + 1, 11 * n + h, 32, 2)
    queries.append((3 * n + h, 31)); queries.append((11 * n + h, 32)); queries.append((5 * n + h, None))
    h = 25000                                      # three keys (the byte index hashes them to six bits): quotients 0, 1, 2
    for q in range(3):
        put(h + q, q * n + h, 33 + q, 2)
        queries.append((q * n + h, 33 + q))
    queries.append((6 * n + h, None))              # 6 % 6 = 0: its bit is set -> a candidate the walk refutes
    queries.append((9 * n + h, None))              # 9 % 6 = 3: bit clear -> a certain miss
    h = n - 3                                      # the run that ends with the stream: found ones are found, a miss runs off
    put(h, 1 * n + h, 26, 2); put(h + 1, 2 * n + h, 27, 2); put(h + 2, 0 * n + h + 2, 28, 2)
    queries.append((2 * n + h, 27)); queries.append((0 * n + h + 2, 28))
    run_off_query = (9 * n + h + 1, None)
    torch.cuda.synchronize()
    host = torch.empty(24 + n * 24, dtype=torch.uint8)
    host[:24] = torch.frombuffer(bytearray(struct.pack("<qqq", n, 24, 1)), dtype=torch.uint8)
    host[24:].view(torch.int32).view(n, 6).copy_(rec)
    img = host.numpy()
    monkeypatch.setenv("KG_PARTITION", "1")
    with hp.SignatureTable.from_device_ptr(rec.data_ptr(), n, 0, keepalive=rec) as tab:
        for with_run_off in (False, True):
            qs = queries + ([run_off_query] if with_run_off else [])
            # every query k-mer twice in a protein of its own, 30 copies of the set so that the batch has some bulk
            prots = [(K.decode(v) + "A" + K.decode(v) + "AA").encode() for v, _ in qs] * 30
            off = np.zeros(len(prots) + 1, dtype=np.int64)
            np.cumsum([len(p) for p in prots], out=off[1:])
            sb = b"".join(prots)
            ora0 = oracle.run(img, sb, off, aa=True, lookup_mode=0, min_hits=2)
            ora1 = oracle.run(img, sb, off, aa=True, lookup_mode=1, min_hits=2)
            assert ora0["hits"].tobytes() == ora1["hits"].tobytes()
            want = {}
            for k, (v, oi) in enumerate(qs):
                got = sorted({int(x["oI"]) for x in ora0["hits"][ora0["hits"]["container"] == k]})
                assert got == ([oi] if oi is not None else []), (k, v, got, oi)
            assert bool(ora0["lookup_aborted"]) == with_run_off
            for bidx, counters in (("1", False), ("1", True), ("0", False)):
                monkeypatch.setenv("KG_BIDX", bidx)
                with tab.scan(sb, off, hp.Params(aa=True, min_hits=2, counters=counters)) as r:
                    assert r.stats["partitioned"] == 1
                    # 4: the byte home index, 1: the tags (with and without counters)
                    assert r.stats["part_levels"] == (4 if bidx == "1" and not counters else 1)
                    assert r.hits().tobytes() == ora0["hits"].tobytes(), (bidx, counters, with_run_off)
                    assert r.calls().tobytes() == ora0["calls"].tobytes() and r.otu().tobytes() == ora0["otu"].tobytes()
                    assert r.stats["lookup_ran_off"] == int(with_run_off), (bidx, counters, r.stats["lookup_ran_off"])
    del rec, host
    torch.cuda.empty_cache()
