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


def _restore(recs_by_rank, idx_by_rank, n_total_seqs: int, per: int, dtype):
    """Concatenate per-rank records (container ids local to the rank's shard, grouped by container in
    ascending order) into global container order.  Vectorised: no per-sequence Python loop."""
    n_cont = n_total_seqs * per
    gcont, recs = [], []
    for r, rec in enumerate(recs_by_rank):
        if len(rec) == 0:
            continue
        lc = rec["container"].astype(np.int64)
        gcont.append(idx_by_rank[r][lc // per] * per + lc % per)
        recs.append(rec)
    starts = np.zeros(n_cont + 1, dtype=np.int64)
    if not recs:
        return np.zeros(0, dtype=dtype), starts
    gcont = np.concatenate(gcont)
    rec = np.concatenate(recs)
    order = np.argsort(gcont, kind="stable")          # stable: keeps the emission order inside a container
    rec = rec[order]
    rec["container"] = gcont[order].astype(np.uint32)
    np.cumsum(np.bincount(gcont, minlength=n_cont), out=starts[1:])
    return rec, starts


def gather_records(local: dict, shard_idx: np.ndarray, n_total_seqs: int, per: int, device=None) -> Optional[dict]:
    """Gather per-rank results to rank 0 and restore the original sequence order.

    local: {"calls": CALL records, "otu": OTU records (one per local sequence), optional "hits"} as numpy
    arrays with container ids local to the shard (container // per = local sequence index).
    Returns on rank 0 {"calls", "container_call_start", "otu" [, "hits", "container_hit_start"]} in global
    numbering, None elsewhere."""
    from . import _native as N
    device = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")

    def tobytes(a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8).copy()).to(device)

    names = ["calls", "otu"] + (["hits"] if "hits" in local else [])
    parts = {"idx": _gather_var(tobytes(np.asarray(shard_idx, dtype=np.int64)))}
    for nm in names:
        parts[nm] = _gather_var(tobytes(local[nm]))
    if dist.get_rank() != 0:
        return None
    dt = {"calls": N.CALL_DTYPE, "otu": N.OTU_DTYPE, "hits": N.HIT_DTYPE, "idx": np.dtype("<i8")}
    world = dist.get_world_size()
    dec = {nm: [np.frombuffer(parts[nm][r].cpu().numpy().tobytes(), dtype=dt[nm]) for r in range(world)] for nm in parts}
    otu = np.zeros(n_total_seqs, dtype=N.OTU_DTYPE)
    seen = np.zeros(n_total_seqs, dtype=np.int64)
    for r in range(world):
        otu[dec["idx"][r]] = dec["otu"][r]
        seen[dec["idx"][r]] += 1
    assert (seen == 1).all(), "every sequence must belong to exactly one rank"
    out = {"otu": otu}
    out["calls"], out["container_call_start"] = _restore(dec["calls"], dec["idx"], n_total_seqs, per, N.CALL_DTYPE)
    if "hits" in local:
        out["hits"], out["container_hit_start"] = _restore(dec["hits"], dec["idx"], n_total_seqs, per, N.HIT_DTYPE)
    return out
