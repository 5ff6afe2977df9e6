"""sanitizers on the CPU side (GPU AddressSanitizer is not available on the pool).  The C oracle is built
with -fsanitize=address,undefined (oracle/Makefile, libkgoracle_asan.so) and run in a child process -- libasan has to be
the first library of the process, so it is preloaded there -- over the known-answer cases and fuzz workloads; the native
front end's FASTA reader (tests/native/fasta_harness.cpp) gets the same treatment as an executable."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_env(**extra):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"       # (CPython itself leaks at exit)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env.update(extra)
    return env


def test_oracle_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libkgoracle_asan.so"], check=True, stdout=subprocess.DEVNULL)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "no libasan in this image"
    env = _san_env(LD_PRELOAD=libasan, KGO_LIB_PATH=os.path.join(ROOT, "oracle", "libkgoracle_asan.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_worker.py"), "30"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "sanitizer worker ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]


def test_fasta_reader_under_asan_and_ubsan(tmp_path):
    import test_native_fasta as F
    exe = str(tmp_path / "fasta_harness_san")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "native", "fasta_harness.cpp"), "-lz"], check=True)
    for name in sorted(F.CASES):
        rng = np.random.default_rng(sum(map(ord, name)))
        text = F.CASES[name](rng)
        path = tmp_path / "q.fa"
        path.write_bytes(text.encode("latin-1"))
        for threads in (1, 7):
            r = subprocess.run([exe, str(path)], capture_output=True, text=True, env=_san_env(KG_FASTA_THREADS=str(threads)))
            assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (name, r.stderr[-3000:])
            assert r.stdout == F._mirror(text), name
