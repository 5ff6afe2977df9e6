"""ctypes binding of oracle/libkgoracle.so -- TEST INFRASTRUCTURE ONLY (see kg_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# KGO_LIB_PATH: another build of the same source (tests/test_sanitizers.py loads oracle/libkgoracle_asan.so in a child process)
LIB_PATH = os.environ.get("KGO_LIB_PATH") or os.path.join(HERE, "libkgoracle.so")

HIT_DTYPE = np.dtype([("container", "<u4"), ("from0InProt", "<i4"), ("oI", "<i4"),
                      ("avgOffFromEnd", "<i4"), ("fI", "<i4"), ("functionWt", "<f4")])
CALL_DTYPE = np.dtype([("container", "<u4"), ("start", "<i4"), ("end", "<i4"), ("count", "<i4"),
                       ("fI", "<i4"), ("weightedHits", "<f4")])
OTU_DTYPE = np.dtype([("n", "<i4"), ("count", "<i4", (5,)), ("oI", "<i4", (5,))])


class Params(C.Structure):
    _fields_ = [("aa", C.c_int32), ("order_constraint", C.c_int32), ("min_hits", C.c_int32),
                ("min_weighted_hits", C.c_int32), ("max_gap", C.c_int32), ("reserved", C.c_int32),
                ("input_size_limit", C.c_int64)]


class Result(C.Structure):
    _fields_ = [("n_seqs", C.c_int64), ("n_containers", C.c_int64), ("n_hits", C.c_int64), ("n_calls", C.c_int64),
                ("hits", C.c_void_p), ("container_hit_start", C.c_void_p), ("calls", C.c_void_p),
                ("container_call_start", C.c_void_p), ("otu", C.c_void_p),
                ("residues", C.c_int64), ("windows_valid", C.c_int64), ("slots_inspected", C.c_int64),
                ("t_prepare", C.c_double), ("t_lookup", C.c_double), ("t_group", C.c_double),
                ("lookup_aborted", C.c_int32), ("hit_events", C.c_void_p), ("container_tail_events", C.c_void_p),
                ("n_processed", C.c_int32), ("processed_tenth", C.c_int32 * 64), ("processed_found", C.c_int64 * 64),
                ("kmers_found", C.c_int64), ("skip_failed_bytes", C.c_int64), ("read_eof", C.c_int32)]


_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, "kg_oracle.c"), os.path.join(HERE, "kg_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", HERE, "-B", os.path.basename(LIB_PATH)], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.kgo_run.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_int, C.POINTER(Result)]
        lib.kgo_result_free.argtypes = [C.POINTER(Result)]
        lib.kgo_result_free.restype = None
        lib.kgo_last_error.restype = C.c_char_p
        lib.kgo_encoded_kmer.argtypes = [C.c_void_p, C.c_int64]
        lib.kgo_encoded_kmer.restype = C.c_int64
        lib.kgo_translate.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.kgo_translate.restype = None
        lib.kgo_rev_comp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.kgo_rev_comp.restype = None
        lib.kgo_gather_hits.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_int64]
        lib.kgo_gather_hits.restype = C.c_int64
        lib.kgo_format_java_f.argtypes = [C.c_float, C.c_int, C.c_char_p, C.c_size_t]
        _lib = lib
    return _lib


def _view(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


def run(table_image, seq, offsets, aa=False, order_constraint=False, min_hits=5, min_weighted_hits=0,
        max_gap=200, lookup_mode=0, input_size_limit=20_000_000) -> dict:
    """Whole reference path on the CPU.  table_image: bytes/ndarray of the uncompressed file."""
    lib = load()
    tab = np.frombuffer(table_image, dtype=np.uint8) if not isinstance(table_image, np.ndarray) else table_image
    tab = np.ascontiguousarray(tab.view(np.uint8).reshape(-1))
    s = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    s = np.ascontiguousarray(s.view(np.uint8).reshape(-1))
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    p = Params(int(aa), int(order_constraint), int(min_hits), int(min_weighted_hits), int(max_gap), 0,
               int(input_size_limit))
    r = Result()
    rc = lib.kgo_run(tab.ctypes.data, tab.nbytes, C.byref(p), s.ctypes.data if s.size else None,
                     off.ctypes.data, off.size - 1, lookup_mode, C.byref(r))
    if rc != 0:
        raise RuntimeError("oracle: " + lib.kgo_last_error().decode())
    out = {
        "hits": _view(r.hits, r.n_hits, HIT_DTYPE),
        "container_hit_start": _view(r.container_hit_start, r.n_containers + 1, np.dtype("<i8")),
        "calls": _view(r.calls, r.n_calls, CALL_DTYPE),
        "container_call_start": _view(r.container_call_start, r.n_containers + 1, np.dtype("<i8")),
        "otu": _view(r.otu, r.n_seqs, OTU_DTYPE),
        "hit_events": _view(r.hit_events, r.n_hits, np.dtype("u1")),
        "container_tail_events": _view(r.container_tail_events, r.n_containers, np.dtype("u1")),
        "residues": r.residues, "windows_valid": r.windows_valid, "slots_inspected": r.slots_inspected,
        "t_prepare": r.t_prepare, "t_lookup": r.t_lookup, "t_group": r.t_group,
        "lookup_aborted": bool(r.lookup_aborted),
        # literal merge-join only (lookup_mode 0): the "Processed: NN%" lines (tenth, found-so-far) in print order, kmersFound,
        # and how the table stream failed (KGJ:1016-1049)
        "processed": [(int(r.processed_tenth[i]), int(r.processed_found[i])) for i in range(r.n_processed)],
        "kmers_found": int(r.kmers_found), "skip_failed_bytes": int(r.skip_failed_bytes), "read_eof": bool(r.read_eof),
    }
    lib.kgo_result_free(C.byref(r))
    return out


def gather_hits(hits: np.ndarray, container=0, otu=None, **kw):
    """gatherHits + processSetOfHits on one container.  Returns (calls, otu_record)."""
    lib = load()
    p = Params(int(kw.get("aa", 0)), int(kw.get("order_constraint", 0)), int(kw.get("min_hits", 5)),
               int(kw.get("min_weighted_hits", 0)), int(kw.get("max_gap", 200)), 0, 20_000_000)
    h = np.ascontiguousarray(hits.astype(HIT_DTYPE))
    o = np.zeros(1, dtype=OTU_DTYPE) if otu is None else np.ascontiguousarray(otu.copy())
    cap = max(16, len(h))
    calls = np.zeros(cap, dtype=CALL_DTYPE)
    n = lib.kgo_gather_hits(C.byref(p), h.ctypes.data, len(h), container, o.ctypes.data, calls.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("oracle: reference crash path")
    return calls[:n].copy(), o


def format_java_f(v: float, precision: int = 6) -> str:
    buf = C.create_string_buffer(64)
    load().kgo_format_java_f(C.c_float(v), precision, buf, 64)
    return buf.value.decode()
