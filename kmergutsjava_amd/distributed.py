"""Multi-GPU layer: one process per GPU, contigs sharded over ranks, the read-only signature table
replicated, per-rank record buffers gathered to rank 0 with torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference has no counterpart (single thread, KGJ = lib/src/kmergutsjava/KmerGutsJava.java);
the sharding is legal because every sequence is independent: hits depend only on the sequence's
own k-mers and the table, and the aggregation state is per sequence (KGJ:528, 540).  The exchange
step is therefore one gather of variable-length record buffers at the end; there is no
all-reduce.

The exchange (gather_records): one gather of the buffer sizes to rank 0, then ONE group of point-to-point
transfers towards rank 0 -- only rank 0 allocates receive buffers (exact sizes, no padding), the
senders hand over the library's own device buffers (ScanResult.device_view: no host hop on the RCCL
path).  CALL / OTU records are a few KB and are put back in FASTA order on the host; hit records
(24 B each, ~110 MB per rank for a 1 Gbp batch on 8 GPUs) stay in HBM: rank 0 scatters every
rank's records to their final place in the global (container, from0InProt) order with index
arithmetic on the device -- the per-rank lists are already ordered, so no sort is needed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_sequences(lengths: Sequence[int], world_size: int, weights: Optional[Sequence[float]] = None) -> List[np.ndarray]:
    """Greedy longest-first balancing of whole sequences over ranks (LPT).  Returns, per rank, the
    ascending list of sequence indices it owns.  Deterministic; ties go to the lowest rank.
    weights: relative share of the work per rank (default: equal) -- the rank that gathers everybody's records and puts
    them in order can be given less to scan."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if world_size == 1:
        return [np.arange(len(lengths), dtype=np.int64)]
    wt = np.ones(world_size) if weights is None else np.asarray(weights, dtype=np.float64)
    assert wt.shape == (world_size,) and (wt > 0).all()
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.float64)
    owner = np.empty(len(lengths), dtype=np.int64)
    for i in order:
        cost = float(lengths[i]) + 24.0            # a small per-sequence cost keeps empty sequences spread
        r = int(np.argmin((load + cost) / wt))
        owner[i] = r
        load[r] += cost
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


def _as_u8(x, device) -> torch.Tensor:
    """numpy array / torch tensor -> contiguous 1-D uint8 tensor on `device` (no copy when it already is one)."""
    if isinstance(x, torch.Tensor):
        t = x.contiguous().reshape(-1)
        if t.dtype != torch.uint8:
            t = t.view(torch.uint8)
        return t if t.device == torch.device(device) else t.to(device)
    a = np.ascontiguousarray(x)
    t = torch.from_numpy(a.reshape(-1).view(np.uint8).copy() if a.size else np.zeros(0, np.uint8))
    return t.to(device)


class _Pending:
    """The transfers of one gather_buffers_start call: requests, receive buffers (rank dst) and whatever must stay alive
    until they are done (the senders' buffers belong to the library's result object)."""

    def __init__(self, reqs, out, dev, keep):
        self.reqs, self.out, self.dev, self.keep = reqs, out, dev, keep

    def wait(self):
        """On return the payload has arrived (dst) / has left (senders).  With RCCL a request's wait() makes torch's current
        stream wait for the transfer; the stream is then synchronised so that the senders' buffers may be freed."""
        for req in self.reqs:
            req.wait()
        self.reqs = []
        if self.dev.type == "cuda":
            torch.cuda.current_stream(self.dev).synchronize()
        self.keep = None
        return self.out


def gather_buffers_start(bufs: List[torch.Tensor], dst: int = 0, keep=None) -> _Pending:
    """First half of gather_buffers: the sizes travel in one gather towards dst -- only dst needs them, and only dst pays
    the host round trip that reads them -- and the payload transfers are POSTED (one batch of point-to-point transfers, so
    only dst allocates); the caller goes on (e.g. into the next scan) and collects with .wait()."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = bufs[0].device
    mine = torch.tensor([b.numel() for b in bufs], dtype=torch.int64, device=dev)
    ops, out = [], None
    if rank == dst:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, parts, dst=dst)
        sizes = torch.stack(parts).cpu().tolist()            # [world][K]
        out = [[None] * world for _ in bufs]
        for r in range(world):
            for k, b in enumerate(bufs):
                if r == dst:
                    out[k][r] = b
                else:
                    out[k][r] = torch.empty(sizes[r][k], dtype=torch.uint8, device=dev)
                    if sizes[r][k]:
                        ops.append(dist.P2POp(dist.irecv, out[k][r], r))
    else:
        dist.gather(mine, None, dst=dst)
        for b in bufs:
            if b.numel():
                ops.append(dist.P2POp(dist.isend, b, dst))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return _Pending(reqs, out, dev, (bufs, keep))


def gather_buffers(bufs: List[torch.Tensor], dst: int = 0) -> Optional[List[List[torch.Tensor]]]:
    """Gather K 1-D uint8 buffers of rank-dependent length to rank dst (all ranks call, same K).
    Returns on dst: out[k][r] = buffer k of rank r (its own buffers are passed through, not copied); None elsewhere."""
    return gather_buffers_start(bufs, dst).wait()


def _restore_small(recs_by_rank, idx_by_rank, n_total_seqs: int, per: int, dtype):
    """Concatenate per-rank records (container ids local to the rank's shard, grouped by container in
    ascending order) into global container order.  Host side, for the small record kinds (CALLs)."""
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


def restore_hits(hits_by_rank: List[torch.Tensor], chs_by_rank: List[torch.Tensor], idx_by_rank: List[torch.Tensor],
                 n_total_seqs: int, per: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-rank hit buffers (uint8, 24 B records ordered by local container and position) + their
    container_hit_start (int64) + the ranks' sequence indices -> (hits int32[n, 6], container_hit_start int64) in
    global numbering, on the device the buffers are on.  The records of one sequence are one contiguous piece of its
    rank's buffer (its containers are adjacent) and one contiguous piece of the result, so the work is a segmented copy:
    per rank one shift per local sequence, spread over its records, and one index_copy_ (nothing waits for the host)."""
    dev = hits_by_rank[0].device
    n_cont = n_total_seqs * per
    counts = torch.zeros(n_cont, dtype=torch.int64, device=dev)
    ar = torch.arange(per, dtype=torch.int64, device=dev)
    for chs, idx in zip(chs_by_rank, idx_by_rank):
        if idx.numel():
            g = (idx[:, None] * per + ar[None, :]).reshape(-1)          # global id of every local container
            counts[g] = chs[1:] - chs[:-1]
    starts = torch.zeros(n_cont + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=starts[1:])
    total = sum(int(hb.numel()) // 24 for hb in hits_by_rank)           # (sizes are known on the host: no read-back)
    out = torch.empty((total, 6), dtype=torch.int32, device=dev)
    keep = []
    for hb, chs, idx in zip(hits_by_rank, chs_by_rank, idx_by_rank):
        h = hb.view(torch.int32).view(-1, 6)
        n = h.shape[0]
        if n == 0:
            continue
        seq_lo = chs[:-1:per]                                            # first record of every local sequence
        k = torch.arange(idx.numel(), dtype=torch.int64, device=dev)
        if dev.type == "cuda":
            # one kernel of the library per rank (segmented copy, 24 B in / 24 B out): the torch formulation below needs
            # five passes over the records -- 2.6 ms for the 36.7 M records of a 1 Gbp batch gathered from 8 ranks, as
            # long as the scan of a 125 Mbp shard (tools/restore_time.py)
            from . import _native as N
            seq_first = chs[::per].contiguous()                          # n_local_seqs + 1 entries
            dst_first = starts[idx * per].contiguous()
            cshift = ((idx - k) * per).to(torch.int32).contiguous()
            src = hb if hb.is_contiguous() else hb.contiguous()
            N.check(N.load().kg_restore_hits_device(dev.index if dev.index is not None else torch.cuda.current_device(), src.data_ptr(), n,
                                                    seq_first.data_ptr(), idx.numel(), dst_first.data_ptr(), cshift.data_ptr(),
                                                    out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            keep.append((src, seq_first, dst_first, cshift))             # alive until the kernel has run (stream order frees them safely)
            continue
        seq_n = chs[per::per] - seq_lo
        to = torch.repeat_interleave(starts[idx * per] - seq_lo, seq_n, output_size=n)       # where a record moves to
        cshift = torch.repeat_interleave(((idx - k) * per).to(torch.int32), seq_n, output_size=n)
        hh = h.clone()
        hh[:, 0] += cshift                                               # container ids < 2^31 (kg_scan's own limit)
        out.index_copy_(0, torch.arange(n, dtype=torch.int64, device=dev) + to, hh)
    return out, starts


class RecordExchange:
    """One exchange step in two halves (exchange_start / .finish), so that a caller can run the next scan while the
    record buffers of this one travel: start posts the transfers (rank 0 learns the sizes and posts the receives), finish
    waits for them and, on rank 0, puts the records back in the original sequence order."""

    def __init__(self, pending, with_hits, n_total_seqs, per):
        self.pending, self.with_hits, self.n_total_seqs, self.per = pending, with_hits, n_total_seqs, per

    def finish(self) -> Optional[dict]:
        """Rank 0: the records in global order.  The hit records are restored by work ENQUEUED on torch's current stream
        (kg_restore_hits_device, slicing) that reads the per-rank buffers -- rank 0's own are zero-copy views of the
        library's result -- so the caller must let that stream run (synchronise it, or wait on an event) before it closes
        the ScanResult the views belong to: the library's block cache takes blocks back as idle."""
        from . import _native as N
        got = self.pending.wait()
        self.pending = None
        if got is None:
            return None
        n_total_seqs, per = self.n_total_seqs, self.per
        world = dist.get_world_size()
        idx_t = [b.view(torch.int64) for b in got[0]]
        idx = [t.cpu().numpy() for t in idx_t]
        calls = [np.frombuffer(b.cpu().numpy().tobytes(), dtype=N.CALL_DTYPE) for b in got[1]]
        otus = [np.frombuffer(b.cpu().numpy().tobytes(), dtype=N.OTU_DTYPE) for b in got[2]]
        otu = np.zeros(n_total_seqs, dtype=N.OTU_DTYPE)
        seen = np.zeros(n_total_seqs, dtype=np.int64)
        for r in range(world):
            otu[idx[r]] = otus[r]
            seen[idx[r]] += 1
        assert (seen == 1).all(), "every sequence must belong to exactly one rank"
        out = {"otu": otu}
        out["calls"], out["container_call_start"] = _restore_small(calls, idx, n_total_seqs, per, N.CALL_DTYPE)
        if self.with_hits:
            out["hits"], out["container_hit_start"] = restore_hits(got[3], [b.view(torch.int64) for b in got[4]], idx_t,
                                                                   n_total_seqs, per)
        return out


def exchange_start(local: Dict[str, object], shard_idx: np.ndarray, n_total_seqs: int, per: int, device=None, keep=None) -> RecordExchange:
    """First half of gather_records (same arguments).  `keep`: an object that owns the buffers in `local` (the
    ScanResult) and must not be freed before .finish() has returned."""
    device = torch.device(device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    with_hits = "hits" in local
    idx_np = np.ascontiguousarray(np.asarray(shard_idx, dtype=np.int64))
    bufs = [_as_u8(idx_np, device), _as_u8(local["calls"], device), _as_u8(local["otu"], device)]
    if with_hits:
        chs = local.get("container_hit_start")
        if chs is None:                                  # numpy records: offsets from the (sorted) container column
            cont = np.asarray(local["hits"]["container"], dtype=np.int64)
            chs = np.searchsorted(cont, np.arange(len(idx_np) * per + 1))
        bufs += [_as_u8(local["hits"], device), _as_u8(chs if isinstance(chs, torch.Tensor) else np.asarray(chs, dtype=np.int64), device)]
    return RecordExchange(gather_buffers_start(bufs, 0, keep=(local, keep)), with_hits, n_total_seqs, per)


def gather_records(local: Dict[str, object], shard_idx: np.ndarray, n_total_seqs: int, per: int, device=None) -> Optional[dict]:
    """Gather per-rank results to rank 0 and restore the original sequence order.

    local: {"calls": CALL records, "otu": OTU records (one per local sequence)} and optionally
    {"hits": hit records, "container_hit_start": int64[n_local_containers + 1]}; numpy arrays or torch tensors
    (ScanResult.device_view: the library's own HBM buffers), container ids local to the shard
    (container // per = local sequence index).  "container_hit_start" may be omitted for numpy hit records.
    Returns on rank 0 {"calls", "container_call_start", "otu"} as numpy arrays [+ "hits" (int32[n, 6] tensor on the
    exchange device; .cpu().numpy().view(HIT_DTYPE) gives records) and "container_hit_start" (int64 tensor)] in
    global numbering, None elsewhere."""
    return exchange_start(local, shard_idx, n_total_seqs, per, device).finish()
