"""The native front end's FASTA reader (mmap, cut in front of caption lines, one thread per piece) returns the records
of the sequential reader -- the Python mirror's restatement of readFasta (KGJ:1132-1192) -- on well-formed, ragged
and malformed inputs.  CPU only: the reader is compiled into a small harness without the GPU library."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("h") / "fasta_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "fasta_harness.cpp"),
                    "-lz"], check=True)
    return exe


def _fnv(b: bytes) -> int:
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _mirror(text: str) -> str:
    from kmergutsjava_amd.kmer_guts_java import read_fasta
    out = []
    try:
        read_fasta(text, lambda n, s, d: out.append("%s\t%d\t%016x\n" % (n, len(s), _fnv(s.encode("latin-1")))))
    except (ValueError, IndexError) as ex:
        return "ERROR\t%s\n" % ex
    return "".join(out)


def _random_fasta(rng, n_rec, eol="\n"):
    parts = []
    for k in range(n_rec):
        parts.append("%s>rec%d %s%s" % (" " * int(rng.integers(0, 2)), k, "descr \t x" if k % 3 else "", eol))
        for _ in range(int(rng.integers(1, 6))):
            ln = "".join(rng.choice(list("ACGTNacgt *"), size=int(rng.integers(0, 70))))
            parts.append(ln + eol)
        if k % 5 == 0:
            parts.append(eol)
        if k % 17 == 0:
            parts.append(">" + eol)              # a bare '>' ends the record and is skipped (KGJ:1163-1180)
    return "".join(parts)


CASES = {
    "lf": lambda rng: _random_fasta(rng, 400),
    "crlf": lambda rng: _random_fasta(rng, 300, "\r\n"),
    "cr_only": lambda rng: _random_fasta(rng, 50, "\r"),
    "junk_before_first_caption": lambda rng: "x\n\nAC\n" + _random_fasta(rng, 20),
    "wrong_caption_in_the_middle": lambda rng: _random_fasta(rng, 200) + "\n>\nnot a caption\n" + _random_fasta(rng, 200),
    "caption_without_sequence_at_end": lambda rng: _random_fasta(rng, 100) + ">last\n",
    "caption_followed_by_caption": lambda rng: _random_fasta(rng, 100) + ">a\n>b\nACGT\n" + _random_fasta(rng, 100),
    "no_trailing_newline": lambda rng: _random_fasta(rng, 64) + ">z\nACGT",
    "empty": lambda rng: "",
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("threads", [1, 7])
def test_threaded_reader_equals_sequential(harness, tmp_path, name, threads):
    rng = np.random.default_rng(sum(map(ord, name)))
    text = CASES[name](rng)
    want = _mirror(text)
    path = tmp_path / "q.fa"
    path.write_bytes(text.encode("latin-1"))
    got = subprocess.run([harness, str(path)], check=True, capture_output=True, text=True,
                         env=dict(os.environ, KG_FASTA_THREADS=str(threads))).stdout
    if want.startswith("ERROR"):
        assert got == want
    else:
        assert got == want and (name == "empty" or got.count("\n") > 10)
