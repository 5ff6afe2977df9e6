"""Python face of the C ABI: the signature table resident in HBM and one scan of a batch.

Replaces, for a batch of sequences, the reference's prepareQuery/addKmers -> sort -> lookup ->
gatherHits/processSetOfHits chain (KGJ:1051-1074, 900-922, 1076-1095, 944-1034, 385-514;
"KGJ:n" = reference lib/src/kmergutsjava/KmerGutsJava.java line n).  All arithmetic happens in
libkmerguts_hip.so on the GPU; this file only moves pointers.
"""
from __future__ import annotations

import ctypes as C
import weakref
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
    progress: bool = False          # KG_F_PROGRESS: ScanResult.progress() / .hit_slots() (the "Processed: NN%" lines, KGJ:1016-1025)

    def to_native(self) -> N.KgParams:
        flags = ((N.KG_F_COUNTERS if self.counters else 0) | (N.KG_F_SKIP_AGGREGATE if self.skip_aggregate else 0) |
                 (N.KG_F_PROGRESS if self.progress else 0))
        return N.KgParams(int(self.aa), int(self.order_constraint), int(self.min_hits),
                          int(self.min_weighted_hits), int(self.max_gap), flags)


class _DevMem:
    """`count` items of `typestr` at a raw HBM address, exposed through __cuda_array_interface__ (version 2)."""

    def __init__(self, ptr: int, count: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2,
                                         "strides": None}
        self._owner = owner


def device_tensor(ptr: int, count: int, typestr: str = "|u1", owner=None, device: Optional[int] = None):
    """torch tensor over library-owned device memory (no copy).  `owner` is kept alive by the tensor."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    if count == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8 if typestr == "|u1" else torch.int64, device=dev)
    t = torch.as_tensor(_DevMem(ptr, count, typestr, owner), device=dev)
    t._kg_owner = owner
    return t


class ScanResult:
    """Owns one kg_result.  Record arrays are numpy structured arrays (copies)."""

    def __init__(self, handle: int, table=None):
        self._h = C.c_void_p(handle)
        self._table = table              # a result must be freed before its table (kg_result_free uses the table's caches)
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

    def copy_hits(self, first: int = 0, count: Optional[int] = None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """hits[first : first + count] into caller-owned memory (a new array unless `out` is given)."""
        lib = self._need()
        count = self.stats["n_hits"] - first if count is None else count
        if out is None:
            out = np.empty(count, dtype=N.HIT_DTYPE)
        assert out.dtype == N.HIT_DTYPE and out.flags.c_contiguous and len(out) >= count
        N.check(lib.kg_result_copy_hits(self._h, first, count, out.ctypes.data if count else None))
        return out[:count]

    def container_hit_start(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_hit_start(self._h), self.stats["n_containers"] + 1, np.dtype("<i8"))

    def calls(self, copy: bool = True) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_calls(self._h), self.stats["n_calls"], N.CALL_DTYPE, None if copy else self)

    def container_call_start(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_call_start(self._h), self.stats["n_containers"] + 1, np.dtype("<i8"))

    def otu(self, copy: bool = True) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_otu(self._h), self.stats["n_seqs"], N.OTU_DTYPE, None if copy else self)

    def hit_events(self) -> np.ndarray:
        """One KG_EV_* byte per hit record: what gatherHits did there (for the -d stream)."""
        lib = self._need()
        return N.view(lib.kg_result_hit_events(self._h), self.stats["n_hits"], np.dtype("u1"))

    def container_tail_events(self) -> np.ndarray:
        lib = self._need()
        return N.view(lib.kg_result_container_tail_events(self._h), self.stats["n_containers"], np.dtype("u1"))

    def hit_slots(self) -> np.ndarray:
        """KG_F_PROGRESS scans: the table slot every hit record was found at (uint32, parallel to hits())."""
        lib = self._need()
        ptr = lib.kg_result_hit_slots(self._h)
        if not ptr and self.stats["n_hits"]:
            raise N.KmerGutsNativeError(-1, (lib.kg_last_error() or b"").decode())
        return N.view(ptr, self.stats["n_hits"], np.dtype("<u4"))

    def progress(self) -> dict:
        """KG_F_PROGRESS scans: kg_progress as a dict (first slot visited per tenth of the table, last slot visited, first
        home slot behind the end of a short table stream, whether a walk ran off its end, records in the stream)."""
        lib = self._need()
        p = N.KgProgress()
        N.check(lib.kg_result_progress(self._h, C.byref(p)))
        return p.as_dict()

    def device_hits_ptr(self) -> int:
        return self._need().kg_result_device_hits(self._h) or 0

    def device_calls_ptr(self) -> int:
        return self._need().kg_result_device_calls(self._h) or 0

    def device_otu_ptr(self) -> int:
        return self._need().kg_result_device_otu(self._h) or 0

    def device_view(self, what: str):
        """Zero-copy torch view (uint8, or int64 for the two start arrays) of a record array where the library left
        it in HBM: "hits", "calls", "otu", "container_hit_start", "container_call_start".  Valid until close();
        plumbing for callers that keep working on the device (RCCL gather, on-device comparisons)."""
        lib = self._need()
        st = self.stats
        n_cont = st["n_containers"]
        spec = {"hits": (lib.kg_result_device_hits, st["n_hits"] * N.HIT_DTYPE.itemsize, "|u1"),
                "calls": (lib.kg_result_device_calls, st["n_calls"] * N.CALL_DTYPE.itemsize, "|u1"),
                "otu": (lib.kg_result_device_otu, st["n_seqs"] * N.OTU_DTYPE.itemsize, "|u1"),
                "container_hit_start": (lib.kg_result_device_container_hit_start, n_cont + 1, "<i8"),
                "container_call_start": (lib.kg_result_device_container_call_start, n_cont + 1, "<i8")}[what]
        return device_tensor(spec[0](self._h) or 0, spec[1], spec[2], owner=self)

    def subset(self, seq_idx, events: bool = False) -> dict:
        """The records of the sequences seq_idx (ascending indices into the batch), renumbered as if those sequences
        had been scanned as a batch of their own: legal because every sequence is independent (hits depend on its own
        k-mers and the table only, the aggregation state is per sequence, KGJ:528, 540).  The hit records are sliced
        where they are (HBM) and only the slices come to the host."""
        import torch
        st = self.stats
        idx = np.asarray(seq_idx, dtype=np.int64)
        per = st["n_containers"] // st["n_seqs"] if st["n_seqs"] else 1
        chs, ccs = self.container_hit_start(), self.container_call_start()
        cont = (idx[:, None] * per + np.arange(per)[None, :]).reshape(-1)               # old container ids, in order
        h_n = chs[cont + 1] - chs[cont]
        c_n = ccs[cont + 1] - ccs[cont]
        new_chs = np.zeros(len(cont) + 1, dtype=np.int64); np.cumsum(h_n, out=new_chs[1:])
        new_ccs = np.zeros(len(cont) + 1, dtype=np.int64); np.cumsum(c_n, out=new_ccs[1:])
        dv = self.device_view("hits").view(torch.int32).view(-1, 6)
        # the containers of one sequence are adjacent in hits[]: one slice per sequence
        parts = [dv[int(chs[i * per]):int(chs[i * per + per])] for i in idx]
        hits = (torch.cat(parts).cpu().numpy() if parts else np.zeros((0, 6), np.int32)).reshape(-1).view(N.HIT_DTYPE).copy()
        hits["container"] = np.repeat(np.arange(len(cont), dtype=np.uint32), h_n)
        allc = self.calls()
        calls = (np.concatenate([allc[int(ccs[i * per]):int(ccs[i * per + per])] for i in idx]) if len(idx)
                 else np.zeros(0, N.CALL_DTYPE))
        calls["container"] = np.repeat(np.arange(len(cont), dtype=np.uint32), c_n)
        out = {"hits": hits, "container_hit_start": new_chs, "calls": calls, "container_call_start": new_ccs,
               "otu": self.otu()[idx]}
        if events:
            ev = self.hit_events()
            out["hit_events"] = (np.concatenate([ev[int(chs[i * per]):int(chs[i * per + per])] for i in idx]) if len(idx)
                                 else np.zeros(0, np.uint8))
            out["container_tail_events"] = self.container_tail_events()[cont]
        return out

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


def aggregate_hits(hits: np.ndarray, container_hit_start, n_seqs: int, params: Optional[Params] = None,
                   otu_init: Optional[np.ndarray] = None, device: int = 0) -> ScanResult:
    """gatherHits / processSetOfHits / the OTU buffer (KGJ:385-524) on the GPU over caller-held hit records (HIT_DTYPE,
    ordered by container and from0InProt).  otu_init: OTU_DTYPE[n_seqs], the oICounts buffers to start from."""
    params = params or Params()
    p = params.to_native()
    h = np.ascontiguousarray(hits, dtype=N.HIT_DTYPE)
    chs = np.ascontiguousarray(np.asarray(container_hit_start, dtype=np.int64))
    if chs.size != n_seqs * (1 if params.aa else 6) + 1:
        raise ValueError("container_hit_start must have n_seqs * (1 or 6) + 1 entries")
    init = None if otu_init is None else np.ascontiguousarray(otu_init, dtype=N.OTU_DTYPE)
    out = C.c_void_p()
    N.check(N.load().kg_aggregate_hits(device, C.byref(p), h.ctypes.data if h.size else None, chs.ctypes.data, n_seqs,
                                       init.ctypes.data if init is not None and init.size else None, C.byref(out)))
    return ScanResult(out.value)


class SignatureTable:
    """kmer.table.mem_map resident on one GPU (KGJ:749-753, 924-942)."""

    def __init__(self, handle: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keep = keepalive
        self._results = weakref.WeakSet()      # results still open: closed with the table, before it

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

    def live_device_bytes(self) -> int:
        """Device bytes of scratch / result blocks handed out by the table's block cache and not yet returned."""
        return int(N.load().kg_table_live_device_bytes(self._h))

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
        r = ScanResult(out.value, self)
        self._results.add(r)
        return r

    def close(self) -> None:
        if self._h:
            for r in list(self._results):
                r.close()
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
