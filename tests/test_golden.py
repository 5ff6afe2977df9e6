"""Committed golden vectors (tests/golden/vectors_r01.json, made by tests/golden/make_golden.py from the
pure-Python restatement -- not reference output, parity unpinned) against the C oracle (CPU) and the
HIP path + host text layer (GPU)."""
import base64
import io
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _vectors():
    return json.load(open(os.path.join(HERE, "golden", "vectors_r01.json")))["vectors"]


def _inputs(v):
    from kmergutsjava_amd.kmer_guts_java import read_fasta
    ids, seqs = [], []
    read_fasta(v["fasta"], lambda n, s, d: (ids.append(n), seqs.append(s.encode())))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    return ids, b"".join(seqs), off, base64.b64decode(v["table_b64"])


def _check_records(v, hits, calls, otu):
    want_h = np.array([tuple(h) for h in v["hits"]], dtype=hits.dtype) if v["hits"] else np.zeros(0, hits.dtype)
    want_c = np.array([tuple(c) for c in v["calls"]], dtype=calls.dtype) if v["calls"] else np.zeros(0, calls.dtype)
    assert hits.tobytes() == want_h.tobytes(), v["name"]
    assert calls.tobytes() == want_c.tobytes(), v["name"]
    for s, want in enumerate(v["otu"]):
        got = [[int(otu[s]["count"][j]), int(otu[s]["oI"][j])] for j in range(int(otu[s]["n"]))]
        assert got == want, (v["name"], s)


@pytest.mark.parametrize("idx", range(5))
def test_c_oracle_reproduces_golden(oracle, idx):
    v = _vectors()[idx]
    ids, sb, off, img = _inputs(v)
    for mode in (0, 1):
        o = oracle.run(img, sb, off, aa=v["aa"], lookup_mode=mode, **v["params"])
        _check_records(v, o["hits"], o["calls"], o["otu"])
    assert len(v["calls"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(5))
def test_hip_and_cli_reproduce_golden(idx, tmp_path):
    from kmergutsjava_amd import hotpath, KmerGutsJava
    v = _vectors()[idx]
    ids, sb, off, img = _inputs(v)
    p = v["params"]
    with hotpath.SignatureTable.from_bytes(img) as tab:
        with tab.scan(sb, off, hotpath.Params(aa=v["aa"], **p)) as r:
            _check_records(v, r.hits(), r.calls(), r.otu())
    # the whole drop-in path: data directory + FASTA file -> report text, through KmerGutsJava.main
    d = tmp_path / "data"
    d.mkdir()
    (d / "kmer.table.mem_map").write_bytes(img)
    (d / "function.index").write_text("".join("%d\t%s\n" % (i, f) for i, f in enumerate(v["functions"])))
    (tmp_path / "q.fa").write_text(v["fasta"])
    args = ["-D", str(d), "-q", str(tmp_path / "q.fa"), "-o", str(tmp_path / "out.txt")]
    if v["aa"]:
        args.append("-a")
    if p.get("order_constraint"):
        args.append("-O")
    for flag, key in (("-m", "min_hits"), ("-g", "max_gap"), ("-M", "min_weighted_hits")):
        if key in p:
            args += [flag, str(p[key])]
    KmerGutsJava.main(args)
    assert (tmp_path / "out.txt").read_text() == v["report"]
    # the native front end (C++ over the same C ABI) prints the same bytes
    import subprocess
    from kmergutsjava_amd import build
    cli = build.build_cli()
    args[args.index("-o") + 1] = str(tmp_path / "out_cli.txt")
    subprocess.run([cli] + args, check=True, stdout=subprocess.DEVNULL)
    assert (tmp_path / "out_cli.txt").read_text() == v["report"]
