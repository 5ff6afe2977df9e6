"""Host-side behaviour of the library around a scan: one scan at a time per table (KG_ERR_BUSY instead of shared streams
and pinned words being corrupted), and nothing left behind by a scan that fails in the middle (KG_TEST_FAIL_ALLOC makes
the n-th device allocation of a call fail; kg_table_live_device_bytes must be 0 afterwards and the next scan normal)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload():
    from kmergutsjava_amd import synth
    rec, placed, keys = synth.random_table(3_000_017, 0.5, 31)
    img = synth.table_image(rec)
    seq, off = synth.dna_uniform_config(40, 500_000, 33)
    return img, seq.numpy(), off


def test_a_second_concurrent_scan_on_one_table_is_turned_away(monkeypatch):
    from kmergutsjava_amd import hotpath, _native as N
    img, sb, off = _workload()
    monkeypatch.setenv("KG_PARTITION", "1")
    with hotpath.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off, hotpath.Params()) as r0:
            want = r0.hits().tobytes()
        codes, results = [], []
        start = threading.Barrier(4)

        def worker():
            start.wait()
            try:
                with tab.scan(sb, off, hotpath.Params()) as r:       # ctypes releases the GIL for the call
                    results.append(r.hits().tobytes())
                    codes.append(0)
            except N.KmerGutsNativeError as e:
                codes.append(e.code)
        for _ in range(3):                                           # three rounds of four simultaneous callers
            ts = [threading.Thread(target=worker) for _ in range(4)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        assert set(codes) <= {0, N.KG_ERR_BUSY}, codes
        assert codes.count(0) >= 3 and N.KG_ERR_BUSY in codes, codes       # one winner per round at least; some turned away
        assert all(x == want for x in results)
        assert tab.live_device_bytes() == 0


@pytest.mark.parametrize("mode", ["0", "1", "2"])          # direct, partitioned on the byte home index, partitioned on the tags
def test_a_scan_that_fails_in_the_middle_leaves_nothing_behind(monkeypatch, mode):
    from kmergutsjava_amd import hotpath, _native as N
    img, sb, off = _workload()
    monkeypatch.setenv("KG_PARTITION", "0" if mode == "0" else "1")
    monkeypatch.setenv("KG_BIDX", "0" if mode == "2" else "1")
    with hotpath.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off, hotpath.Params()) as r0:
            want = (r0.hits().tobytes(), r0.calls().tobytes())
            assert tab.live_device_bytes() > 0                       # the open result's record arrays
        assert tab.live_device_bytes() == 0
        failed = 0
        for n in range(1, 60):
            monkeypatch.setenv("KG_TEST_FAIL_ALLOC", str(n))
            try:
                with tab.scan(sb, off, hotpath.Params()) as r:
                    assert (r.hits().tobytes(), r.calls().tobytes()) == want
                break                                                # the call makes fewer than n allocations: done
            except N.KmerGutsNativeError as e:
                assert e.code == N.KG_ERR_NOMEM, e
                failed += 1
                assert tab.live_device_bytes() == 0, "allocation %d failed and %d bytes stayed live" % (n, tab.live_device_bytes())
        monkeypatch.delenv("KG_TEST_FAIL_ALLOC")
        assert failed >= 15
        with tab.scan(sb, off, hotpath.Params()) as r:
            assert (r.hits().tobytes(), r.calls().tobytes()) == want


@pytest.mark.parametrize("bidx", ["1", "0"])
def test_resize_and_rerun_hands_every_list_block_back(monkeypatch, bidx):
    """KG_TEST_TINY_LISTS starts the hit / candidate lists (and with them the two ordering buffers) at one chunk: the
    attempt is thrown away and redone with the exact sizes.  Every block of the first attempt has to be back in the
    cache once the result is closed (round 3 leaked the ordering buffers of the first attempt until kg_table_close)."""
    from kmergutsjava_amd import hotpath
    img, sb, off = _workload()
    monkeypatch.setenv("KG_PARTITION", "1")
    monkeypatch.setenv("KG_BIDX", bidx)
    with hotpath.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off, hotpath.Params()) as r0:
            want = (r0.hits().tobytes(), r0.calls().tobytes())
        assert tab.live_device_bytes() == 0
        monkeypatch.setenv("KG_TEST_TINY_LISTS", "1")
        for _ in range(2):
            with tab.scan(sb, off, hotpath.Params()) as r:
                assert r.stats["scan_launches"] >= 2, r.stats            # the resize-and-rerun path did run
                assert (r.hits().tobytes(), r.calls().tobytes()) == want
            assert tab.live_device_bytes() == 0, "%d bytes stayed live after a resized scan" % tab.live_device_bytes()
