"""The two CPU restatements (C oracle, literal merge-join and direct probing; pure-Python model)
agree on random inputs, and the product's host-side text layer (FASTA reader, report writer,
Java %f) reproduces the model's report from the oracle's binary records."""
import io

import numpy as np
import pytest
import torch

from oracle import kgj_model as M


def _case(seed, dna, n_seqs, n_kmers, num_sigs=8009, n_sigs=2500):
    from kmergutsjava_amd import synth
    seq, off, rec, keys = synth.high_density_config(n_seqs, n_kmers, num_sigs, n_sigs, seed=seed, dna=dna)
    return synth.table_image(rec), seq.numpy().tobytes(), off


def _fasta(sb, off, width=60, blank=True):
    out = []
    for k in range(len(off) - 1):
        s = sb[off[k]:off[k + 1]].decode()
        out.append(">seq%d some description\t here\n" % k)
        out += [s[i:i + width] + "\n" for i in range(0, len(s), width)]
        if blank:
            out.append("\n")
    return "".join(out)


@pytest.mark.parametrize("dna", [True, False])
@pytest.mark.parametrize("oc", [False, True])
def test_c_oracle_equals_python_model(oracle, dna, oc):
    img, sb, off = _case(601 + int(dna), dna, 5, 60)
    fn = ["function %d" % i for i in range(64)]
    for mh, mw, gap in ((5, 0, 200), (2, 3, 9), (3, 0, 30)):
        o0 = oracle.run(img, sb, off, aa=not dna, lookup_mode=0, min_hits=mh, min_weighted_hits=mw, max_gap=gap,
                        order_constraint=oc)
        o1 = oracle.run(img, sb, off, aa=not dna, lookup_mode=1, min_hits=mh, min_weighted_hits=mw, max_gap=gap,
                        order_constraint=oc)
        for k in ("hits", "calls", "otu", "container_hit_start", "container_call_start"):
            assert o0[k].tobytes() == o1[k].tobytes(), k
        assert o0["lookup_aborted"] == o1["lookup_aborted"]      # merge-join threw <=> a direct probe walked off the end
        m = M.Model(aa=not dna, order_constraint=oc, min_hits=mh, min_weighted_hits=mw, max_gap=gap)
        text = m.run(img, fn, _fasta(sb, off))
        assert (m.info_lines[-1] == "Error: null") == o0["lookup_aborted"]   # KGJ:799-802 vs KGJ:1031-1033
        mh_ = np.array(m.hits, dtype=[("c", "<u4"), ("p", "<i4"), ("o", "<i4"), ("a", "<i4"), ("f", "<i4"), ("w", "<f4")])
        assert mh_.tobytes() == o0["hits"].tobytes()
        mc = np.array(m.calls, dtype=[("c", "<u4"), ("s", "<i4"), ("e", "<i4"), ("n", "<i4"), ("f", "<i4"), ("w", "<f4")]) \
            if m.calls else np.zeros(0, dtype=o0["calls"].dtype)
        assert mc.tobytes() == o0["calls"].tobytes()
        for s, otu in enumerate(m.otus):
            rec = o0["otu"][s]
            assert [(int(rec["count"][j]), int(rec["oI"][j])) for j in range(int(rec["n"]))] == otu
        # host text layer (product code) from the oracle's records == the model's report
        from kmergutsjava_amd.kmer_guts_java import KmerGutsJava, read_fasta
        host = KmerGutsJava.__new__(KmerGutsJava)
        host.aa = not dna
        ids, seqs = [], []
        read_fasta(_fasta(sb, off), lambda n_, s_, d_: (ids.append(n_), seqs.append(s_)))
        assert [len(s) for s in seqs] == list(np.diff(off))
        per = 1 if not dna else 6
        pw = io.StringIO()
        ccs = o0["container_call_start"]
        for s in range(len(ids)):
            calls = [o0["calls"][ccs[s * per + f]:ccs[s * per + f + 1]] for f in range(per)]
            host.write_record(pw, ids[s], len(seqs[s]), calls, o0["otu"][s], fn)
        assert pw.getvalue() == text
        if mh == 5 and not oc:
            assert text.count("CALL\t") > 10


def test_fasta_reader_matches_model_on_odd_input():
    from kmergutsjava_amd.kmer_guts_java import read_fasta, load_indexed_array
    text = ("\n  \n>id1 first  record\twith tabs\nACGT \n acgt\n\n>id2\n\n\nMKV\r\nLLL\r>id3 x\nAA\n>\n>id4\nC\n")
    a, b = [], []
    read_fasta(text, lambda *x: a.append(x))
    M.read_fasta(text, lambda *x: b.append(x))
    assert a == b and [x[0] for x in a] == ["id1", "id2", "id3", "id4"]
    assert a[0][1] == "ACGT  acgt" and a[0][2] == "first record with tabs"
    for bad in (">id1\n>id2\nAC\n", "ACGT\n", ">id1\n"):
        with pytest.raises(ValueError):
            read_fasta(bad, lambda *x: None)
        with pytest.raises(ValueError):
            M.read_fasta(bad, lambda *x: None)
    assert load_indexed_array("0\ta\n1\tb c\n") == ["a", "b c"] == M.load_indexed_array("0\ta\n1\tb c\n")
    with pytest.raises(ValueError):
        load_indexed_array("0\ta\n2\tb\n")


def test_host_static_helpers_match_oracle(oracle):
    """toAminoAcidOff / compl / revComp / dnaChar / encodedKmer of the host mirror (KGJ:111-318)."""
    from kmergutsjava_amd import KmerGutsJava as H
    lib = oracle.load()
    for c in range(256):
        ch = chr(c)
        assert H.toAminoAcidOff(ch) == lib.kgo_to_amino_acid_off(c) == M.to_amino_acid_off(ch)
        assert ord(H.compl(ch)) == lib.kgo_compl(c) == ord(M.compl(ch))
        assert H.dnaChar(ch) == lib.kgo_dna_char(c) == M.dna_char(ch)
    assert H.revComp("ACGTNacgtnSs") == M.rev_comp("ACGTNacgtnSs") == "SSnacgtNACGT"
    assert H.encodedKmer([12] * 8, 0) == M.encoded_kmer([12] * 8, 0)
    assert H.MAX_ENCODED == 20 ** 8 and H.K == 8 and H().status()["version"] == "0.0.1"


def test_literal_batches_do_not_change_results(oracle):
    """Results are independent of the <= inputSizeLimit batching of the literal lookup."""
    img, sb, off = _case(611, True, 12, 40)
    a = oracle.run(img, sb, off, lookup_mode=0, input_size_limit=20_000_000)
    b = oracle.run(img, sb, off, lookup_mode=0, input_size_limit=500)
    for k in ("hits", "calls", "otu"):
        assert a[k].tobytes() == b[k].tobytes()


def test_reference_crash_paths_are_refused(oracle):
    img, sb, off = _case(612, False, 2, 10)
    with pytest.raises(RuntimeError):
        oracle.run(img, sb, off, aa=True, min_hits=1)
