"""World-size-2 rehearsal of the multi-GPU layer on CPU (gloo): shard whole sequences over ranks,
scan each shard independently (the oracle stands in for the GPU here -- this test is about the
sharding and the gather, not the kernels), gather CALL / OTU / hit records to rank 0, restore the
original order, compare with the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker8(rank, world, port, tmp):
    """Eight ranks, the bench's geometry: weighted shards (the gathering rank scans less), a rank with an empty shard's worth
    of hits is possible, exchange in its two halves, restore on rank 0."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kmergutsjava_amd import distributed as kd, synth
    from oracle import kgo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, off, rec, keys = synth.high_density_config(37, 30, 8009, 2500, seed=778, dna=True)
        img = synth.table_image(rec)
        sb = seq.numpy()
        lens = np.diff(off)
        shards = kd.shard_sequences(lens, world, [1.0 - 0.15 * world / 8.0] + [1.0] * (world - 1))     # bench.py's default sink share
        assert sorted(np.concatenate(shards).tolist()) == list(range(len(lens))) and all(len(x) for x in shards)
        mine = shards[rank]
        s_seq, s_off = kd.take_shard(sb, off, mine)
        loc = kgo.run(img, s_seq, s_off, lookup_mode=1)
        local = {k: torch.from_numpy(loc[k].view(np.uint8).copy()) for k in ("calls", "otu", "hits")}
        local["container_hit_start"] = torch.from_numpy(loc["container_hit_start"].copy())
        pending = [kd.exchange_start(local, mine, len(lens), 6, "cpu") for _ in range(2)]      # two exchanges in flight (overlap)
        for ex in pending:
            got = ex.finish()
            if rank == 0:
                whole = kgo.run(img, sb, off, lookup_mode=1)
                for k in ("calls", "container_call_start", "otu"):
                    assert got[k].tobytes() == whole[k].tobytes(), k
                assert got["hits"].numpy().tobytes() == whole["hits"].tobytes()
                assert np.array_equal(got["container_hit_start"].numpy(), whole["container_hit_start"])
                assert len(whole["calls"]) > 20 and len(whole["hits"]) > 500
            else:
                assert got is None
        if rank == 0:
            open(os.path.join(tmp, "ok8"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_world8(tmp_path, oracle):
    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert (tmp_path / "ok8").exists()


def _worker(rank, world, port, dna, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kmergutsjava_amd import distributed as kd, synth
    from oracle import kgo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, off, rec, keys = synth.high_density_config(11, 45, 8009, 2500, seed=777, dna=dna)
        img = synth.table_image(rec)
        sb = seq.numpy()
        per = 6 if dna else 1
        lens = np.diff(off)
        shards = kd.shard_sequences(lens, world)
        assert sorted(np.concatenate(shards).tolist()) == list(range(len(lens)))
        mine = shards[rank]
        s_seq, s_off = kd.take_shard(sb, off, mine)
        loc = kgo.run(img, s_seq, s_off, aa=not dna, lookup_mode=1)
        # records as numpy arrays (offsets derived) and as torch tensors with the offsets handed over, the form
        # ScanResult.device_view gives on the RCCL path
        as_np = {k: loc[k] for k in ("calls", "otu", "hits")}
        as_t = {k: torch.from_numpy(loc[k].view(np.uint8).copy()) for k in ("calls", "otu", "hits")}
        as_t["container_hit_start"] = torch.from_numpy(loc["container_hit_start"].copy())
        for local in (as_np, as_t, {k: loc[k] for k in ("calls", "otu")}):
            got = kd.gather_records(local, mine, len(lens), per, device="cpu")
            if rank == 0:
                whole = kgo.run(img, sb, off, aa=not dna, lookup_mode=1)
                for k in ("calls", "container_call_start", "otu"):
                    assert got[k].tobytes() == whole[k].tobytes(), k
                if "hits" in local:
                    assert got["hits"].numpy().tobytes() == whole["hits"].tobytes()
                    assert np.array_equal(got["container_hit_start"].numpy(), whole["container_hit_start"])
                else:
                    assert "hits" not in got
                assert len(whole["calls"]) > 5 and len(whole["hits"]) > 100
            else:
                assert got is None
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dna", [True, False])
def test_shard_and_gather_world2(tmp_path, dna, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, dna, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_balance():
    from kmergutsjava_amd import distributed as kd, synth
    lens = synth.contig_mix_lengths(50_000_000, 301)
    for w in (2, 4, 8):
        sh = kd.shard_sequences(lens, w)
        loads = np.array([lens[i].sum() for i in sh])
        assert loads.max() / loads.mean() < 1.05, (w, loads)
        assert sum(len(i) for i in sh) == len(lens)
        # the gathering rank can be given a smaller share
        sh = kd.shard_sequences(lens, w, [0.8] + [1.0] * (w - 1))
        loads = np.array([lens[i].sum() for i in sh], dtype=np.float64)
        assert abs(loads[0] / loads[1:].mean() - 0.8) < 0.03 and loads[1:].max() / loads[1:].mean() < 1.05, (w, loads)
        assert sorted(np.concatenate(sh).tolist()) == list(range(len(lens)))
