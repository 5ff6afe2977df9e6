"""Shared helpers of the parity tests."""
from __future__ import annotations

import numpy as np


def assert_same_records(gpu, ora, what=""):
    """gpu: kmergutsjava_amd.hotpath.ScanResult, ora: dict from oracle.kgo.run.  Bit-exact."""
    gh, oh = gpu.hits(), ora["hits"]
    assert gpu.stats["n_hits"] == len(oh), "%s n_hits %d != %d" % (what, gpu.stats["n_hits"], len(oh))
    assert gh.tobytes() == oh.tobytes(), "%s hit records differ (first diff at %s)" % (what, _first_diff(gh, oh))
    assert np.array_equal(gpu.container_hit_start(), ora["container_hit_start"]), what + " container_hit_start"
    gc, oc = gpu.calls(), ora["calls"]
    assert len(gc) == len(oc), "%s n_calls %d != %d" % (what, len(gc), len(oc))
    assert gc.tobytes() == oc.tobytes(), "%s call records differ (first diff at %s)" % (what, _first_diff(gc, oc))
    assert np.array_equal(gpu.container_call_start(), ora["container_call_start"]), what + " container_call_start"
    go, oo = gpu.otu(), ora["otu"]
    assert go.tobytes() == oo.tobytes(), "%s OTU records differ (first diff at %s)" % (what, _first_diff(go, oo))
    ge, oe = gpu.hit_events(), ora["hit_events"]              # what the -d stream shows at every record
    assert np.array_equal(ge, oe), "%s hit events differ at %s" % (what, np.flatnonzero(ge != oe)[:5])
    assert np.array_equal(gpu.container_tail_events(), ora["container_tail_events"]), what + " tail events"
    assert gpu.stats["residues"] == ora["residues"], what + " residues"
    # where the reference's table stream would have thrown EOFException ("Error: null" instead of "Kmers found")
    assert gpu.stats["lookup_ran_off"] == int(ora["lookup_aborted"]), what + " lookup_ran_off"
    import os
    mode = os.environ.get("KG_PARTITION")
    knobs = any(os.environ.get(k) is not None for k in ("KG_PART_OVF_GROUPS", "KG_PART_SLACK"))
    if not knobs:       # 1 = overflow beyond the list, 2 = spin guard of the scatter pass: never without a forcing knob
        assert gpu.stats["fallback"] == 0, "%s: partitioned attempt thrown away (fallback %d)" % (what, gpu.stats["fallback"])
    if mode == "0":
        assert gpu.stats["partitioned"] == 0, what + ": direct strategy requested"
    if mode == "1" and gpu.stats["n_blocks"] > 0 and os.environ.get("KG_PART_OVF_GROUPS") is None and os.environ.get("KG_PART_SLACK") is None and _partition_fits(gpu.stats):
        assert gpu.stats["partitioned"] == 1, what + ": the partitioned strategy fell back to direct probing"
    if gpu.stats["partitioned"] == 1:
        # what the tag pass probed: 4 = the byte home index (scans without KG_F_COUNTERS), 1 = the tags
        want = 4 if os.environ.get("KG_BIDX", "1") != "0" and gpu.stats["windows_valid"] < 0 else 1
        assert gpu.stats["part_levels"] == want, "%s: partition levels %d, wanted %d" % (what, gpu.stats["part_levels"], want)


def _partition_fits(stats) -> bool:
    """The library's own eligibility rule for the partitioned strategy (kmerguts_hip.hip, scan_impl)."""
    num_sigs = stats["table_bytes"] // 24
    qmax = 20 ** 8 // num_sigs + 1
    shift = 21
    while shift > 4 and qmax >= (1 << (32 - shift)):
        shift -= 1
    while ((num_sigs + (1 << shift) - 1) >> shift) > 1024:
        shift += 1
    enc = 720 * 16 + 512 if stats["n_containers"] != stats["n_seqs"] else 80 * 16 + 256        # WaveLds<AA> x 16 + tables
    while enc + ((num_sigs + (1 << shift) - 1) >> shift) * 140 > 160 * 1024:
        shift += 1
    return shift < 32 and qmax < (1 << (32 - shift)) and stats["n_blocks"] <= (1 << 23) and 64 <= num_sigs < (1 << 31)


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i].tobytes() != b[i].tobytes():
            return "%d: %s vs %s" % (i, a[i], b[i])
    return "length %d vs %d" % (len(a), len(b))


def plant(seq_bytes: bytes, off, keys, every=40, dna=True, start=10):
    """Overwrite stretches of the sequences with decoded signature k-mers so that hits occur."""
    from kmergutsjava_amd import synth
    s = bytearray(seq_bytes)
    ki = 0
    for k in range(len(off) - 1):
        a, b = int(off[k]), int(off[k + 1])
        p = a + start
        span = 24 if dna else 8
        while p + span <= b and ki < len(keys):
            pep = synth.decode_kmer(int(keys[ki]))
            word = synth.back_translate(pep) if dna else pep
            s[p:p + span] = word.encode()
            ki += 1
            p += every
    return bytes(s)


def chunk_seq_ranges(off, want=4, dna=True, min_chunk_blocks=None):
    """The sequence ranges [a, b) of the chunks the partitioned scan cuts a batch into (kmerguts_hip.hip, scan_impl:
    chunk c starts at the first sequence whose first window block is >= nblocks * c / want; the number of chunks by
    the batch's size -- round(sqrt(blocks / 325 000)), at least two, one below 450 000 blocks -- or, with
    KG_PART_MIN_CHUNK_BLOCKS = min_chunk_blocks, fewer chunks while a chunk would hold fewer blocks than that).
    A DNA block is 192 forward positions, a protein block 64 windows."""
    L = np.asarray(off[1:] - off[:-1], dtype=np.int64)
    nb = (np.maximum(L - 23, 0) + 191) // 192 if dna else (np.maximum(L - 8, 0) + 63) // 64
    ibase = np.zeros(len(L) + 1, dtype=np.int64)
    np.cumsum(nb, out=ibase[1:])
    nblocks = int(ibase[-1])
    if min_chunk_blocks is not None:
        while want > 1 and nblocks // want < min_chunk_blocks:
            want -= 1
    else:
        by_size = 1 if nblocks < 450000 else max(2, int(np.floor(np.sqrt(nblocks / 325000.0) + 0.5)))
        want = min(want, max(1, by_size))
    cuts, clo = [0], [0]
    for c in range(1, want):
        k = int(np.searchsorted(ibase, nblocks * c // want, side="left"))
        cut = int(ibase[k])
        if clo[-1] < cut < nblocks:
            clo.append(cut)
            cuts.append(k)
    cuts.append(len(L))
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
