"""ctypes binding of libkmerguts_hip.so (C ABI: include/kmerguts_hip.h).

The library is the product path.  If it is missing or cannot be loaded this module raises:
there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KG_LIB_PATH") or os.path.join(HERE, "libkmerguts_hip.so")   # KG_LIB_PATH: tuning builds only

KG_OK = 0
KG_ERR_NOMEM = -5
KG_ERR_BUSY = -8
KG_F_COUNTERS = 1
KG_F_SKIP_AGGREGATE = 2
KG_F_PROGRESS = 4
KG_OI_BUFSZ = 5

# every symbol include/kmerguts_hip.h declares
EXPORTS = (
    "kg_table_open", "kg_table_from_memory", "kg_table_from_device", "kg_table_info", "kg_table_live_device_bytes", "kg_table_close",
    "kg_scan", "kg_scan_device", "kg_aggregate_hits", "kg_process_set_of_hits", "kg_result_stats", "kg_result_hits", "kg_result_container_hit_start",
    "kg_result_calls", "kg_result_container_call_start", "kg_result_otu", "kg_result_hit_events",
    "kg_result_container_tail_events", "kg_result_copy_hits", "kg_result_hit_slots", "kg_result_progress", "kg_result_device_hits", "kg_result_device_calls", "kg_result_device_otu",
    "kg_result_device_container_hit_start", "kg_result_device_container_call_start", "kg_result_free", "kg_restore_hits_device",
    "kg_last_error", "kg_version",
)

# event bits (include/kmerguts_hip.h KG_EV_*)
EV_ACCEPTED, EV_RESET_BEFORE, EV_CALL_BEFORE, EV_KEEP2_BEFORE = 0x01, 0x02, 0x04, 0x08
EV_RESET_AFTER, EV_CALL_AFTER, EV_KEEP2_AFTER = 0x10, 0x20, 0x40
EV_TAIL_CALL = 0x01

HIT_DTYPE = np.dtype([("container", "<u4"), ("from0InProt", "<i4"), ("oI", "<i4"),
                      ("avgOffFromEnd", "<i4"), ("fI", "<i4"), ("functionWt", "<f4")])
CALL_DTYPE = np.dtype([("container", "<u4"), ("start", "<i4"), ("end", "<i4"), ("count", "<i4"),
                       ("fI", "<i4"), ("weightedHits", "<f4")])
OTU_DTYPE = np.dtype([("n", "<i4"), ("count", "<i4", (KG_OI_BUFSZ,)), ("oI", "<i4", (KG_OI_BUFSZ,))])
assert HIT_DTYPE.itemsize == 24 and CALL_DTYPE.itemsize == 24 and OTU_DTYPE.itemsize == 44


class KgParams(C.Structure):
    _fields_ = [("aa", C.c_int32), ("order_constraint", C.c_int32), ("min_hits", C.c_int32),
                ("min_weighted_hits", C.c_int32), ("max_gap", C.c_int32), ("flags", C.c_uint32)]


class KgStats(C.Structure):
    _fields_ = [("n_seqs", C.c_int64), ("n_containers", C.c_int64), ("n_blocks", C.c_int64),
                ("n_hits", C.c_int64), ("n_calls", C.c_int64), ("residues", C.c_int64),
                ("windows", C.c_int64), ("windows_valid", C.c_int64), ("slots_inspected", C.c_int64),
                ("table_bytes", C.c_int64), ("ms_scan", C.c_float), ("ms_order", C.c_float),
                ("ms_aggregate", C.c_float), ("ms_total", C.c_float), ("scan_launches", C.c_int32),
                ("partitioned", C.c_int32), ("ms_part_scatter", C.c_float), ("ms_part_tag", C.c_float),
                ("ms_part_verify", C.c_float), ("fallback", C.c_int32), ("part_chunks", C.c_int32),
                ("part_buckets", C.c_int32), ("part_shift", C.c_int32), ("lookup_ran_off", C.c_int32),
                ("agg_pieces", C.c_int32), ("part_levels", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class KgProgress(C.Structure):
    """struct kg_progress (KG_F_PROGRESS scans): what the reference's table stream would have reported (KGJ:1016-1049)."""
    _fields_ = [("first_visited", C.c_int64 * 11), ("last_visited", C.c_int64), ("first_beyond", C.c_int64),
                ("walk_ran_off", C.c_int64), ("stream_slots", C.c_int64), ("found_upto", C.c_int64 * 11), ("kmers_found", C.c_int64)]

    def as_dict(self):
        return {"first_visited": [int(x) for x in self.first_visited], "last_visited": int(self.last_visited),
                "first_beyond": int(self.first_beyond), "walk_ran_off": int(self.walk_ran_off), "stream_slots": int(self.stream_slots),
                "found_upto": [int(x) for x in self.found_upto], "kmers_found": int(self.kmers_found)}


class KmerGutsNativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libkmerguts_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # libkmerguts_hip.so is linked against the system's libamdhip64; PyTorch ships a copy of its own under the same
    # SONAME.  Whichever is loaded first serves both, and torch.cuda only comes up on its own copy (measured on the GPU
    # box: library first, torch second -> torch.cuda.is_available() is False).  The package uses torch for device
    # tensors and torch.distributed, so torch's runtime goes first; KG_NO_TORCH_PRELOAD=1 skips the ~1.5 s import for
    # processes that never touch torch.cuda.
    if not os.environ.get("KG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m kmergutsjava_amd.build` "
            "(hipcc --offload-arch=gfx950).  kmergutsjava_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
    lib.kg_table_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.kg_table_from_memory.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.kg_table_from_device.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(vp)]
    lib.kg_table_info.argtypes = [vp, i64p, i64p, i64p, i64p]
    lib.kg_table_live_device_bytes.argtypes = [vp]
    lib.kg_table_live_device_bytes.restype = C.c_int64
    lib.kg_table_close.argtypes = [vp]
    lib.kg_table_close.restype = None
    lib.kg_scan.argtypes = [vp, C.POINTER(KgParams), vp, vp, C.c_int64, C.POINTER(vp)]
    lib.kg_scan_device.argtypes = [vp, C.POINTER(KgParams), vp, vp, C.c_int64, C.POINTER(vp)]
    lib.kg_aggregate_hits.argtypes = [C.c_int, C.POINTER(KgParams), vp, vp, C.c_int64, vp, C.POINTER(vp)]
    i32p = C.POINTER(C.c_int32)
    lib.kg_process_set_of_hits.argtypes = [C.c_int, C.POINTER(KgParams), vp, C.c_int32, C.c_int32, vp, vp, i32p, i32p, i32p]
    lib.kg_result_stats.argtypes = [vp, C.POINTER(KgStats)]
    for name in ("kg_result_hits", "kg_result_container_hit_start", "kg_result_calls",
                 "kg_result_container_call_start", "kg_result_otu", "kg_result_hit_events",
                 "kg_result_container_tail_events", "kg_result_device_hits", "kg_result_device_calls", "kg_result_device_otu",
                 "kg_result_device_container_hit_start", "kg_result_device_container_call_start"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = vp
    lib.kg_result_copy_hits.argtypes = [vp, C.c_int64, C.c_int64, vp]
    lib.kg_result_hit_slots.argtypes = [vp]
    lib.kg_result_hit_slots.restype = vp
    lib.kg_result_progress.argtypes = [vp, C.POINTER(KgProgress)]
    lib.kg_restore_hits_device.argtypes = [C.c_int, vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp]
    lib.kg_result_free.argtypes = [vp]
    lib.kg_result_free.restype = None
    lib.kg_last_error.restype = C.c_char_p
    lib.kg_version.restype = C.c_char_p
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError if a declared symbol is not exported
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != KG_OK:
        raise KmerGutsNativeError(rc, load().kg_last_error().decode("utf-8", "replace"))


def view(ptr: int, count: int, dtype: np.dtype, owner=None) -> np.ndarray:
    """`count` records at `ptr` (library-owned pinned host memory) as a numpy array.  With `owner` the array is a
    zero-copy read-only view that keeps `owner` (the ScanResult) alive; without, a private copy."""
    if count == 0:
        return np.zeros(0, dtype=dtype)
    if not ptr:
        raise KmerGutsNativeError(-1, load().kg_last_error().decode("utf-8", "replace"))
    buf = (C.c_uint8 * (count * dtype.itemsize)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=count)
    if owner is None:
        return arr.copy()
    buf._owner = owner                      # the ctypes buffer is the array's base: keeps the result alive
    arr.flags.writeable = False
    return arr
