"""Multi-GPU layer: one process per GPU, contigs sharded over ranks, the read-only signature table
replicated, CALL / OTU-COUNTS records gathered to rank 0 with torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference has no counterpart (single thread, KGJ = lib/src/kmergutsjava/KmerGutsJava.java);
the sharding is legal because every sequence is independent: hits depend only on the sequence's
own k-mers and the table, and the aggregation state is per sequence (KGJ:528, 540).  The exchange
step is therefore one gather of variable-length record buffers at the end; there is no
all-reduce.  Hit records (needed only for the -d debug stream) stay sharded in HBM unless
gather_records is asked for them.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_sequences(lengths: Sequence[int], world_size: int) -> List[np.ndarray]:
    """Greedy longest-first balancing of whole sequences over ranks (LPT).  Returns, per rank, the
    ascending list of sequence indices it owns.  Deterministic; ties go to the lowest rank."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += int(lengths[i]) + 24           # a small per-sequence cost keeps empty sequences spread
    return [np.flatnonzero(owner == r) for r in range(world_size)]


def take_shard(seq: np.ndarray, offsets: np.ndarray, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate the sequences `idx` out of a batch."""
    lens = (offsets[1:] - offsets[:-1])[idx]
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    out = np.empty(int(off[-1]), dtype=np.uint8)
    for k, i in enumerate(idx):
        out[off[k]:off[k + 1]] = seq[offsets[i]:offsets[i + 1]]
    return out, off


def _gather_var(t: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather 1-D uint8 tensors of different lengths to rank dst (all ranks call)."""
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=t.device)
    pad[:t.numel()] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)                    # tiny buffers; all_gather exists on every backend
    if dist.get_rank() != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


def gather_records(local: dict, shard_idx: np.ndarray, n_total_seqs: int, per: int, device=None) -> Optional[dict]:
    """Gather per-rank results to rank 0 and restore the original sequence order.

    local: {"calls": CALL records, "container_call_start": int64[n_local*per+1], "otu": OTU records,
            optional "hits" + "container_hit_start"} as numpy arrays, container ids local to the shard.
    Returns on rank 0 the same dict in global numbering, None elsewhere."""
    from . import _native as N
    device = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")

    def tobytes(a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8).copy()).to(device)

    names = ["calls", "container_call_start", "otu"] + (["hits", "container_hit_start"] if "hits" in local else [])
    parts = {"idx": _gather_var(tobytes(np.asarray(shard_idx, dtype=np.int64)))}
    for nm in names:
        parts[nm] = _gather_var(tobytes(local[nm]))
    if dist.get_rank() != 0:
        return None
    dt = {"calls": N.CALL_DTYPE, "otu": N.OTU_DTYPE, "hits": N.HIT_DTYPE,
          "container_call_start": np.dtype("<i8"), "container_hit_start": np.dtype("<i8"), "idx": np.dtype("<i8")}
    world = dist.get_world_size()
    dec = {nm: [np.frombuffer(parts[nm][r].cpu().numpy().tobytes(), dtype=dt[nm]) for r in range(world)]
           for nm in parts}
    otu = np.zeros(n_total_seqs, dtype=N.OTU_DTYPE)
    owner = np.full(n_total_seqs, -1, dtype=np.int64)
    local_of = np.zeros(n_total_seqs, dtype=np.int64)
    for r in range(world):
        idx = dec["idx"][r]
        owner[idx] = r
        local_of[idx] = np.arange(len(idx))
        otu[idx] = dec["otu"][r]
    assert (owner >= 0).all(), "every sequence must belong to exactly one rank"
    out = {"otu": otu}
    for rec, st in (("calls", "container_call_start"),) + ((("hits", "container_hit_start"),) if "hits" in local else ()):
        chunks, starts = [], np.zeros(n_total_seqs * per + 1, dtype=np.int64)
        at = 0
        for s in range(n_total_seqs):
            r, ls = int(owner[s]), int(local_of[s])
            cs = dec[st][r]
            for f in range(per):
                a, b = int(cs[ls * per + f]), int(cs[ls * per + f + 1])
                starts[s * per + f] = at
                if b > a:
                    c = dec[rec][r][a:b].copy()
                    c["container"] = s * per + f
                    chunks.append(c)
                    at += b - a
        starts[-1] = at
        out[rec] = np.concatenate(chunks) if chunks else np.zeros(0, dtype=dt[rec])
        out[st] = starts
    return out
