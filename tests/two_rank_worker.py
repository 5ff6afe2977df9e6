#!/usr/bin/env python3
"""One rank of tests/test_gpu_two_ranks.py (started by torch.distributed.run, world size 2, both ranks on cuda:0, gloo):
the step of bench.py --gpus 2 -- shard the ONE contig list, scan the shard through the HIP library, gather the CALL /
OTU / hit buffers to rank 0, restore the global order -- and, on rank 0, the comparison with the unsharded scan of the
whole list.  Writes <out>/ok on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, total_bp, num_sigs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from kmergutsjava_amd import distributed as kd, hotpath, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    try:
        rec, placed, keys = synth.random_table(num_sigs, 0.5, 202, dev)
        del keys
        tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), num_sigs, 0, keepalive=rec)
        all_lens = synth.contig_mix_lengths(total_bp, 301)
        all_off = synth.offsets_of(all_lens)
        mine = kd.shard_sequences(all_lens, world)[rank]
        lens = all_lens[mine]
        seq = synth.random_dna_at(all_off[mine], lens, 302, dev)
        off = synth.offsets_of(lens)
        torch.cuda.synchronize()
        for with_hits in (True, False):
            with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
                kinds = ("calls", "otu") + (("hits", "container_hit_start") if with_hits else ())
                local = {k: r.device_view(k).cpu() for k in kinds}
                got = kd.gather_records(local, mine, len(all_lens), 6, "cpu")
            if rank == 0:
                whole_seq = synth.random_dna(int(all_off[-1]), 302, dev)
                torch.cuda.synchronize()
                with tab.scan(None, all_off, hotpath.Params(), device_ptr=whole_seq.data_ptr()) as w:
                    assert got["calls"].tobytes() == w.calls().tobytes(), "CALL records differ from the unsharded scan"
                    assert got["otu"].tobytes() == w.otu().tobytes(), "OTU records differ from the unsharded scan"
                    assert np.array_equal(got["container_call_start"], w.container_call_start())
                    if with_hits:
                        assert got["hits"].numpy().tobytes() == w.hits().tobytes(), "hit records differ from the unsharded scan"
                        assert np.array_equal(got["container_hit_start"].numpy(), w.container_hit_start())
                        assert w.stats["n_hits"] > 10000
                    else:
                        assert "hits" not in got
                del whole_seq
            else:
                assert got is None
        dist.barrier()
        tab.close()
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
