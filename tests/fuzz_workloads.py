"""Random small workloads for the differential fuzz tests (tests/test_gpu_fuzz.py) and tools/fuzz_strategies.py:
tables of various sizes and loads, DNA / protein, ragged lengths, planted signatures, low-complexity runs, invalid
characters, parameters, and knobs that force chunking, tiny regions, tiny lists and several regions per hand-out."""
import numpy as np


def workloads(iters, seed):
    from kmergutsjava_amd import synth
    rng = np.random.default_rng(seed)
    for it in range(iters):
        aa = bool(rng.integers(0, 2))
        num_sigs = int(rng.choice([101, 1009, 50021, 200003, 1_000_003, 3_000_017]))
        load = float(rng.uniform(0.2, 0.95))
        n_keys = max(8, int(num_sigs * load))
        keys = synth.random_keys(n_keys, int(rng.integers(1, 1 << 30)))
        rec, placed = synth.build_table(keys, synth.payload_of(keys, int(rng.integers(1, 1 << 30)), n_otu=int(rng.integers(1, 9)),
                                                               n_fn=int(rng.integers(1, 12))), num_sigs)
        img = synth.table_image(rec)
        n_seqs = int(rng.integers(1, 60))
        lens = rng.choice([0, 7, 23, 24, 64, 191, 192, 193, 500, 3000, 20000], size=n_seqs).astype(np.int64)
        if rng.integers(0, 3) == 0:
            lens[int(rng.integers(0, n_seqs))] = int(rng.integers(30000, 150000))
        off = np.zeros(n_seqs + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        gen = synth.random_protein if aa else synth.random_dna
        raw = gen(max(total, 1), int(rng.integers(1, 1 << 30))).numpy()[:total].copy()
        kl = keys.tolist()
        for k in range(n_seqs):                         # plant signatures and low-complexity runs
            a, b = int(off[k]), int(off[k + 1])
            span = 8 if aa else 24
            p = a + int(rng.integers(0, 40))
            step = int(rng.integers(span, 120))
            while p + span <= b and rng.integers(0, 5) != 0:
                pep = synth.decode_kmer(int(kl[int(rng.integers(0, len(kl)))]))
                word = pep if aa else synth.back_translate(pep)
                raw[p:p + span] = np.frombuffer(word.encode(), dtype=np.uint8)
                p += step
            if b - a > 2000 and rng.integers(0, 3) == 0:
                q = a + int(rng.integers(0, b - a - 1500))
                ln = int(rng.integers(300, 1500))
                unit = (b"K", b"KR", b"A")[int(rng.integers(0, 3))] if aa else (b"A", b"AT", b"ACG", b"T")[int(rng.integers(0, 4))]
                raw[q:q + ln] = np.frombuffer((unit * (ln // len(unit) + 1))[:ln], dtype=np.uint8)
            if b - a > 50 and rng.integers(0, 4) == 0:
                raw[a + int(rng.integers(0, b - a))] = ord("X") if aa else ord("N")
        params = dict(aa=aa, order_constraint=bool(rng.integers(0, 2)), min_hits=int(rng.integers(2, 7)),
                      min_weighted_hits=int(rng.integers(0, 4)), max_gap=int(rng.choice([5, 30, 200, 300])))
        env = {"KG_PART_CHUNKS": str(int(rng.integers(1, 6))), "KG_PART_MIN_CHUNK_BLOCKS": "1"}
        if rng.integers(0, 3) == 0:
            env["KG_PART_SLACK"] = str(int(rng.choice([5, 20, 50])))
        if rng.integers(0, 4) == 0:
            env["KG_TEST_TINY_LISTS"] = "1"
        if rng.integers(0, 8) == 0:
            env["KG_PART_OVF_GROUPS"] = str(int(rng.choice([1, 64])))
        if rng.integers(0, 2) == 0:                                  # regions per hand-out of the byte-index pass
            env["KG_INDEX_R"] = str(int(rng.choice([1, 2, 4])))
        # the callers' "2" mode: the tag kernels for every scan (no byte home index)
        env2 = {"KG_BIDX": "0"}
        yield dict(env2=env2, it=it, aa=aa, num_sigs=num_sigs, load=load, img=img, raw=raw, off=off, params=params, env=env)
