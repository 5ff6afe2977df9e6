"""CPU checks of the helpers the BASELINE-size tests and the strong-scaling bench stand on: a shard of whole contigs has
the bases it has in the unsharded batch, the per-chunk sample really draws from every chunk, the chunk mirror follows the
library's cutting rule."""
import numpy as np
import torch

from helpers import chunk_seq_ranges


def test_random_dna_at_reproduces_the_unsharded_batch():
    from kmergutsjava_amd import synth, distributed as kd
    lens = synth.contig_mix_lengths(5_000_000, 301)
    off = synth.offsets_of(lens)
    full = synth.random_dna(int(off[-1]), 302)
    shards = kd.shard_sequences(lens, 3)
    assert sorted(np.concatenate(shards).tolist()) == list(range(len(lens)))
    for mine in shards:
        got = synth.random_dna_at(off[mine], lens[mine], 302, chunk=200_000)         # several pieces
        want = torch.cat([full[int(off[i]):int(off[i + 1])] for i in mine])
        assert torch.equal(got, want)
    assert synth.random_dna_at(off[:0], lens[:0], 302).numel() == 0
    z = np.array([0, 5, 0], dtype=np.int64)                                        # zero-length sequences in the shard
    assert torch.equal(synth.random_dna_at(np.array([7, 100, 9]), z, 302), full[100:105])


def test_spread_sample_and_chunk_ranges():
    from kmergutsjava_amd import synth
    lens = synth.contig_mix_lengths(1_000_000_000, 301)
    off = synth.offsets_of(lens)
    ranges = chunk_seq_ranges(off, 4)
    assert len(ranges) == 4 and ranges[0][0] == 0 and ranges[-1][1] == len(lens)
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    blocks = (np.maximum(lens - 23, 0) + 191) // 192
    per_chunk = [int(blocks[a:b].sum()) for a, b in ranges]
    assert max(per_chunk) - min(per_chunk) < 0.02 * sum(per_chunk)                 # equal work up to one contig
    idx = synth.spread_sample(off, groups=4, per_group=20, max_bp_per_group=2_500_000)
    assert np.all(np.diff(idx) > 0)
    for a, b in ranges:
        got = idx[(idx >= a) & (idx < b)]
        assert len(got) >= 3 and int(lens[got].sum()) <= 2_500_000 + int(lens[got].max())
    # small batches: fewer chunks (two for 125 Mbp, one below ~85 Mbp: the library's size rule), the sample still spreads
    lens2 = synth.contig_mix_lengths(125_000_000, 301)
    off2 = synth.offsets_of(lens2)
    r2 = chunk_seq_ranges(off2, 4)
    assert len(r2) == 2 and r2[0][0] == 0 and r2[0][1] == r2[1][0] and r2[1][1] == len(lens2)
    lens3 = synth.contig_mix_lengths(60_000_000, 301)
    assert chunk_seq_ranges(synth.offsets_of(lens3), 4) == [(0, len(lens3))]
    assert len(chunk_seq_ranges(synth.offsets_of(synth.contig_mix_lengths(500_000_000, 301)), 4)) == 3
    assert chunk_seq_ranges(off2, 4, min_chunk_blocks=600000) == [(0, len(lens2))]      # KG_PART_MIN_CHUNK_BLOCKS: the explicit rule
    idx2 = synth.spread_sample(off2, groups=4, per_group=5)
    assert len(idx2) >= 4 and idx2[0] > 0 and idx2[-1] < len(lens2)
