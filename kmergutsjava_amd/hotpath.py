"""Python face of the C ABI: the signature table resident in HBM and one scan of a batch.

Replaces, for a batch of sequences, the reference's prepareQuery/addKmers -> sort -> lookup ->
gatherHits/processSetOfHits chain (KGJ:1051-1074, 900-922, 1076-1095, 944-1034, 385-514;
"KGJ:n" = reference lib/src/kmergutsjava/KmerGutsJava.java line n).  All arithmetic happens in
libkmerguts_hip.so on the GPU; this file only moves pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _native as N


@dataclass
class Params:
    """The instance fields the hot path reads (KGJ:102-106)."""
    aa: bool = False
    order_constraint: bool = False
    min_hits: int = 5
    min_weighted_hits: int = 0
    max_gap: int = 200
    counters: bool = False
    skip_aggregate: bool = False

    def to_native(self) -> N.KgParams:
        flags = (N.KG_F_COUNTERS if self.counters else 0) | (N.KG_F_SKIP_AGGREGATE if self.skip_aggregate else 0)
        return N.KgParams(int(self.aa), int(self.order_constraint), int(self.min_hits),
                          int(self.min_weighted_hits), int(self.max_gap), flags)


class ScanResult:
    """Owns one kg_result.  Record arrays are numpy structured arrays (copies)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        st = N.KgStats()
        N.check(N.load().kg_result_stats(self._h, C.byref(st)))
        self.stats = st.as_dict()

    def _need(self):
        if not self._h:
            raise ValueError("ScanResult is closed")
        return N.load()

    def hits(self, copy: bool = True) -> np.ndarray:
        """Hit records ordered by (container, from0InProt).  copy=False: zero-copy view of the library's pinned
        buffer, valid until close()."""
        lib = self._need()
        return N.view(lib.kg_result_hits(self._h), self.stats["n_hits"], N.HIT_DTYPE, None if copy else self)

    def container_hit_start(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_hit_start(self._h), self.stats["n_containers"] + 1, np.dtype("<i8"))

    def calls(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_calls(self._h), self.stats["n_calls"], N.CALL_DTYPE)

    def container_call_start(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_call_start(self._h), self.stats["n_containers"] + 1, np.dtype("<i8"))

    def otu(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_otu(self._h), self.stats["n_seqs"], N.OTU_DTYPE)

    def hit_events(self) -> np.ndarray:
        """One KG_EV_* byte per hit record: what gatherHits did there (for the -d stream)."""
        lib = self._need()
        return N.view(lib.kg_result_hit_events(self._h), self.stats["n_hits"], np.dtype("u1"))

    def container_tail_events(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_tail_events(self._h), self.stats["n_containers"], np.dtype("u1"))

    def device_hits_ptr(self) -> int:
        return self._need().kg_result_device_hits(self._h) or 0

    def device_calls_ptr(self) -> int:
        return self._need().kg_result_device_calls(self._h) or 0

    def device_otu_ptr(self) -> int:
        return self._need().kg_result_device_otu(self._h) or 0

    def close(self) -> None:
        if self._h:
            N.load().kg_result_free(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SignatureTable:
    """kmer.table.mem_map resident on one GPU (KGJ:749-753, 924-942)."""

    def __init__(self, handle: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keep = keepalive

    @classmethod
    def open(cls, path: str, device: int = 0) -> "SignatureTable":
        out = C.c_void_p()
        N.check(N.load().kg_table_open(path.encode(), device, C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_bytes(cls, image, device: int = 0) -> "SignatureTable":
        """image: bytes-like / uint8 ndarray holding the whole (uncompressed) file."""
        arr = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else image
        arr = np.ascontiguousarray(arr.view(np.uint8).reshape(-1))
        out = C.c_void_p()
        N.check(N.load().kg_table_from_memory(arr.ctypes.data, arr.nbytes, device, C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_device_ptr(cls, ptr: int, num_sigs: int, device: int = 0, keepalive=None) -> "SignatureTable":
        """Adopt num_sigs 24-byte records already in HBM (e.g. a torch tensor's data_ptr())."""
        out = C.c_void_p()
        N.check(N.load().kg_table_from_device(C.c_void_p(ptr), num_sigs, device, C.byref(out)))
        return cls(out.value, keepalive)

    def info(self) -> dict:
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        N.check(N.load().kg_table_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"numSigs": a.value, "entrySize": b.value, "version": c.value, "occupied": d.value}

    def scan(self, seq, offsets, params: Optional[Params] = None, device_ptr: Optional[int] = None) -> ScanResult:
        """seq: bytes / uint8 ndarray with the concatenated raw sequence characters, or None when
        device_ptr gives their address in HBM.  offsets: int64[n_seqs + 1]."""
        if not self._h:
            raise ValueError("SignatureTable is closed")
        params = params or Params()
        p = params.to_native()
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
        if off.ndim != 1 or off.size < 1:
            raise ValueError("offsets must be int64[n_seqs + 1]")
        n = off.size - 1
        out = C.c_void_p()
        lib = N.load()
        if device_ptr is not None:
            N.check(lib.kg_scan_device(self._h, C.byref(p), C.c_void_p(device_ptr), off.ctypes.data, n, C.byref(out)))
        else:
            arr = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
            arr = np.ascontiguousarray(arr.view(np.uint8).reshape(-1))
            if arr.size < int(off[-1]):
                raise ValueError("sequence buffer shorter than offsets[-1]")
            ptr = arr.ctypes.data if arr.size else None
            N.check(lib.kg_scan(self._h, C.byref(p), ptr, off.ctypes.data, n, C.byref(out)))
        return ScanResult(out.value)

    def close(self) -> None:
        if self._h:
            N.load().kg_table_close(self._h)
            self._h = C.c_void_p(None)
            self._keep = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
