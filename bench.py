#!/usr/bin/env python3
"""bench.py -- throughput of the kmer_guts hot path on MI355X.

One "step" = one pass of the hot path (6-frame translate -> base-20 encode -> signature-table
probe -> ordered hit compaction -> CALL / OTU aggregation) over one batch of synthetic contigs
that is already resident in HBM, through the C ABI (kg_scan_device).  Workload = BASELINE.json
configs[2] (== the metric's configuration): 1 Gbp metagenome-like contig mix against a full-size
KmerGuts signature table.  With N ranks (BASELINE.json configs[3]) the SAME
1 Gbp contig list is cut into N shards of whole contigs (strong scaling, the default; every rank
generates only its shard), each rank scans its shard against its own replica of the table, and the
per-rank CALL / OTU / hit buffers are gathered to rank 0 over RCCL inside the timed region, device
buffer to device buffer (--no-gather-hits leaves the hit records sharded and times only the CALL / OTU
exchange).  --scaling weak gives every rank its own 1 Gbp instead.

`--gpus N` with N > 1 outside a launcher (no WORLD_SIZE in the environment) starts the N ranks itself
(`python -m torch.distributed.run`, one process per GPU, 127.0.0.1 rendezvous) BEFORE anything touches
the GPU and exits with the launcher's code; under a launcher, WORLD_SIZE must equal --gpus.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming)
HBM_MEASURED_GBS = 6290.0      # the same guide's measured streaming rate (SURVEY 8d asks for the fraction of both)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--total-bp", type=int, default=1_000_000_000, help="contig bases per GPU")
    ap.add_argument("--num-sigs", type=int, default=1_400_303_159, help="signature table slots (x 24 B)")
    ap.add_argument("--load", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N = 1: skip the BASELINE configs 2 and 5 that are timed (a few seconds) behind the headline")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-threaded CPU row (0 = min(16, cores available); 1 = skip the row)")
    ap.add_argument("--cpu-sample-bp", type=int, default=100_000_000,
                    help="prefix of the contig list the CPU baseline is timed on (BASELINE.md section 3: >= 100 Mbp, "
                         "in batches of <= 20 M query k-mers)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the one --total-bp contig list sharded over the ranks (BASELINE config 4), "
                         "weak = --total-bp per rank")
    ap.add_argument("--gather-hits", dest="gather_hits", action="store_true", default=True,
                    help="N > 1 (default): the per-rank hit buffers are gathered to rank 0 inside the timed steps, with the "
                         "CALL / OTU records (BASELINE config 4: 'RCCL-over-xGMI gather of hit buffers')")
    ap.add_argument("--no-gather-hits", dest="gather_hits", action="store_false",
                    help="N > 1, secondary mode: only the CALL / OTU records (the report's content) travel inside the timed "
                         "steps; the hit records, which only the -d stream prints, stay sharded in HBM and their gather is "
                         "exercised and timed in two extra steps after the timed region ('hits_gather_probe')")
    ap.add_argument("--no-overlap-exchange", dest="overlap_exchange", action="store_false", default=True,
                    help="N > 1: finish every step's exchange before the next scan starts (default: the record buffers of step i "
                         "travel while step i + 1 is scanned)")
    ap.add_argument("--sink-share", type=float, default=-1.0,
                    help="N > 1, strong scaling: rank 0's share of the contigs relative to an equal share (rank 0 also receives every "
                         "rank's records and puts the hit records in global order: ~1 ms per step for 36.7 M records, tools/restore_time.py). "
                         "Default: 1 - 0.15 * N / 8 when the hit records are gathered, else 1")
    ap.add_argument("--exchange-at-world-1", action="store_true",
                    help="rehearsal (tests/test_gpu_rccl_world1.py): under a launcher with ONE rank, still create the process group "
                         "and run every step's exchange through it -- init_process_group('nccl', device_id=...), the size gather and "
                         "rank 0's restore on real RCCL without a second GPU")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    ap.add_argument("--strategy", choices=["auto", "direct", "partitioned"], default="auto",
                    help="scan strategy of the library (KG_PARTITION): auto picks partitioned probing for large inputs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for "
                                                      "rehearsing the multi-rank path on a one-GPU box)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks (nothing has touched the GPU yet: torch.cuda is not initialised)
        import socket
        import subprocess
        port = args.master_port
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] --gpus %d without a launcher: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(cmd)))
        raise SystemExit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to print a line whose n_gpus "
                         "is not the number of ranks that ran" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))

    os.environ["KG_PARTITION"] = {"auto": "2", "direct": "0", "partitioned": "1"}[args.strategy]
    import torch.distributed as dist
    from kmergutsjava_amd import hotpath, synth
    from kmergutsjava_amd import distributed as kd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if "KG_BENCH_DEVICE" in os.environ:               # rehearsal: several ranks share one GPU (gloo only)
        local_rank = int(os.environ["KG_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    multi = world > 1 or (args.exchange_at_world_1 and "WORLD_SIZE" in os.environ)      # the exchange runs
    if multi:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- synthetic table (replicated) and this rank's contig shard, generated in HBM ----
    t0 = time.time()
    rec, placed, keys = synth.random_table(args.num_sigs, args.load, 202, dev)
    del keys
    torch.cuda.synchronize()
    log("[bench] table: %d slots, %d signatures, %.1f GB, built in %.1f s" %
        (args.num_sigs, placed, args.num_sigs * 24 / 1e9, time.time() - t0))
    tab = hotpath.SignatureTable.from_device_ptr(rec.data_ptr(), args.num_sigs, local_rank, keepalive=rec)

    all_lens = synth.contig_mix_lengths(args.total_bp, 301)
    all_off = synth.offsets_of(all_lens)
    strong = args.scaling == "strong" or world == 1
    sink_share = 1.0
    if world == 1:
        mine, n_total = np.arange(len(all_lens), dtype=np.int64), len(all_lens)
        lens, seq = all_lens, synth.random_dna(int(all_off[-1]), 302, dev)
    elif strong:
        # the one contig list, whole contigs balanced over the ranks; a contig has the bases it has in the unsharded batch
        sink_share = args.sink_share if args.sink_share > 0 else (1.0 - 0.15 * world / 8.0 if args.gather_hits else 1.0)
        mine, n_total = kd.shard_sequences(all_lens, world, [sink_share] + [1.0] * (world - 1))[rank], len(all_lens)
        lens = all_lens[mine]
        seq = synth.random_dna_at(all_off[mine], lens, 302, dev)
    else:
        # every rank its own total_bp: same length mix, different bases; global contig k * world + rank
        mine, n_total = np.arange(len(all_lens), dtype=np.int64) * world + rank, len(all_lens) * world
        lens, seq = all_lens, synth.random_dna(int(all_off[-1]), 302, dev, start=rank * args.total_bp)
    off = synth.offsets_of(lens)
    torch.cuda.synchronize()
    log("[bench] rank %d: %d contigs, %d bp (%s)" % (rank, len(lens), int(off[-1]), "one GPU" if world == 1 else "strong" if strong else "weak"))

    params = hotpath.Params()                      # reference defaults: -m 5 -g 200
    gather_dev = comm_dev
    on_gpu = comm_dev.type == "cuda"

    # N > 1: the exchange of step i (per-rank record buffers -> rank 0, straight out of the library's HBM buffers) is
    # posted when scan i ends and collected when scan i + 1 has ended: the buffers travel over xGMI while the next
    # shard is scanned (--no-overlap-exchange collects at once).  Every exchange is finished, and rank 0 has the
    # records of every step in global order, before the timed region ends.
    in_flight = []                                         # [(ScanResult, RecordExchange, with_hits)]
    last = {"hits_gathered": None}                         # hit records rank 0 received in the latest collected step

    restore_ms = []                                        # rank 0: what finish() cost it per step (waits + reordering)

    def collect():
        while in_flight:
            r, ex, with_hits = in_flight.pop(0)
            t_f = time.perf_counter()
            got = ex.finish()              # rank 0: the records in global order (the hit records stay in HBM; their
            if with_hits and rank == 0:    # reordering is enqueued on torch's stream, beside the scan that follows)
                last["hits_gathered"] = int(got["hits"].shape[0])
            # rank 0's own share is a zero-copy view of the library's d_hits / d_chs, and finish() only ENQUEUED the kernels
            # that read it (kg_restore_hits_device + torch slicing, on torch's current stream): the blocks may go back to the
            # table's block cache -- whose rule is "idle when freed" -- only when that work has run.  (It still ran beside
            # the scan that has just ended; the wait is for what is left of ~1 ms.)
            if on_gpu and rank == 0:
                torch.cuda.current_stream().synchronize()
            if rank == 0:
                restore_ms.append((time.perf_counter() - t_f) * 1e3)
            r.close()

    def step(with_hits=False):
        r = tab.scan(None, off, params, device_ptr=seq.data_ptr())
        st = r.stats
        if multi:
            kinds = ("calls", "otu") + (("hits", "container_hit_start") if with_hits else ())
            local = {k: (r.device_view(k) if on_gpu else r.device_view(k).cpu()) for k in kinds}
            ex = kd.exchange_start(local, mine, n_total, 6, gather_dev, keep=r)
            collect()                                      # the previous step's exchange: done by now
            in_flight.append((r, ex, with_hits))
            if not args.overlap_exchange:
                collect()
        else:
            r.calls(copy=False); r.otu(copy=False)         # the records the report needs leave HBM
            r.close()
        return st

    # one instrumented launch: algorithmic bytes per residue, not timed
    with tab.scan(None, off, hotpath.Params(counters=True), device_ptr=seq.data_ptr()) as r:
        cst = r.stats
    residues = cst["residues"]
    p_bar = cst["slots_inspected"] / residues
    h_bar = cst["n_hits"] / residues
    b_alg = 0.5 + 24.0 * p_bar + 24.0 * h_bar

    for _ in range(args.warmup):
        step(args.gather_hits and multi)
    collect()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    del restore_ms[:]
    t1 = time.perf_counter()
    scan_ms, total_ms, agg_ms, order_ms = [], [], [], []
    pass_ms = {"scatter_until_last_chunk": [], "tag_verify_tail": []}
    hits = calls = 0
    partitioned = False
    for _ in range(args.steps):
        st = step(args.gather_hits and multi)
        partitioned = bool(st["partitioned"])
        pass_ms["scatter_until_last_chunk"].append(st["ms_part_scatter"]); pass_ms["tag_verify_tail"].append(st["ms_part_verify"])
        scan_ms.append(st["ms_scan"]); total_ms.append(st["ms_total"])
        agg_ms.append(st["ms_aggregate"]); order_ms.append(st["ms_order"])
        hits, calls = st["n_hits"], st["n_calls"]
        assert st["scan_launches"] == 1, "staging buffer resized inside the timed region"
    collect()                                              # the last step's exchange ends inside the timed region
    barrier()
    elapsed = time.perf_counter() - t1
    hits_gathered_timed = last["hits_gathered"]
    rank0_restore_ms = float(np.mean(restore_ms)) if restore_ms else None

    hits_probe = None
    if multi and not args.gather_hits:
        # outside the timed region: the same step with the per-rank hit buffers gathered to rank 0 as well
        step(True)
        collect()
        barrier()
        t2 = time.perf_counter()
        for _ in range(2):
            step(True)
        collect()
        barrier()
        hits_probe = {"ms_per_step": (time.perf_counter() - t2) / 2 * 1e3}

    tot = torch.tensor([elapsed, float(residues), float(hits)], dtype=torch.float64, device=comm_dev)
    if multi:
        mx = tot[:1].clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot[1:].clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, residues_all, hits_all = float(mx[0]), float(sm[0]), float(sm[1])
    else:
        residues_all, hits_all = float(residues), float(hits)

    if rank == 0:
        ms_scan = float(np.mean(scan_ms))
        achieved = b_alg * residues / (ms_scan * 1e-3) / 1e9          # GB/s, algorithmic bytes / scan-kernel time
        achieved_step = b_alg * residues / (elapsed / args.steps) / 1e9   # ... / whole step (rank 0's bytes, max-over-ranks time)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")        # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                if (tj.get("total_bp") == args.total_bp and tj.get("num_sigs") == args.num_sigs and
                        tj.get("strategy") == ("partitioned" if partitioned else "direct")):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command, FETCH doubled per MI355X_MICROARCH.md; not measured in this run"
                                      % tj.get("round", "?"))
            except Exception:
                traffic = None
        out = {
            "metric": "amino-acid residues/sec scanned (and hits/sec) on 1 Gbp synthetic, 1/2/4/8 MI355X",
            "value": residues_all * args.steps / elapsed,
            "unit": "residues/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": (None if world == 1 else "strong" if strong else "weak"), "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "hits_per_s": hits_all * args.steps / elapsed,
            "config": {"workload": "%.3g Gbp contig mix (0.5 kbp..1.024 Mbp, uniform ACGT) %s, 6-frame translate + "
                                   "8-mer lookup vs %d-slot signature table (%.1f GB, load %.2f), -m 5 -g 200"
                                   % (args.total_bp / 1e9, "per GPU" if not strong else
                                      ("on 1 GPU" if world == 1 else "cut into %d shards of whole contigs, one per GPU" % world),
                                      args.num_sigs, args.num_sigs * 24 / 1e9, args.load),
                       "total_bp_rank0": int(off[-1]), "contigs_rank0": int(len(lens)),
                       "total_bp_all_ranks": int(all_off[-1]) * (1 if strong else world),
                       "num_sigs": args.num_sigs, "residues_rank0": int(residues), "residues_all_ranks": int(residues_all),
                       "hits_rank0": int(hits), "hits_all_ranks": int(hits_all), "calls_rank0": int(calls),
                       "exchange": (None if not multi else "per-rank %s buffers -> rank 0, %s point-to-point, device buffers, %s"
                                    % ("CALL/OTU/hit" if args.gather_hits else "CALL/OTU",
                                       "RCCL" if args.backend == "nccl" else args.backend,
                                       "step i's transfers overlap scan i + 1 (all finished inside the timed region)"
                                       if args.overlap_exchange else "finished before the next scan")),
                       "hits_gathered_rank0": hits_gathered_timed,
                       "hits_gather_probe": (None if hits_probe is None else dict(
                           hits_probe, note="two extra steps after the timed region, hit records (24 B each) gathered to rank 0 "
                                            "too and put in global (container, from0InProt) order on the device",
                           hit_bytes_all_ranks=int(hits_all) * 24)),
                       "sink_share": (sink_share if world > 1 and strong else None),
                       # rank 0's extra work per step, on the host clock: waiting for the step's transfers, CALL / OTU records
                       # back in FASTA order, the hit records' segmented copy into global order (enqueued + waited for)
                       "rank0_restore_ms": rank0_restore_ms,
                       "parallelism": "contig shards x%d, table replicated" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "peak_measured": HBM_MEASURED_GBS, "frac_vs_measured_peak": achieved / HBM_MEASURED_GBS,
                         "achieved_step": achieved_step, "frac_step": achieved_step / HBM_PEAK_GBS,
                         "frac_note": "frac = algorithmic bytes / scan-stage time (HIP events on the library's streams); "
                                      "frac_step = the same bytes / ms_per_step (ordering, aggregation, records to host / "
                                      "exchange included)",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "bound_note": ("priced against HBM as SURVEY 8d prescribes; the partitioned scan moves fewer HBM "
                                        "bytes than the algorithmic count and its passes are bound by VALU/LDS issue "
                                        "(scatter), the L2's line-gather rate (index pass: one byte of the table's home index per query) and random HBM lines (verification): "
                                        "DESIGN.md section 6") if partitioned else
                                       "direct probing: every 16-byte probe moves a 128-byte line from HBM",
                         "kernel": ("scan stage = kg::part_scatter_kernel || kg::bucket_index_kernel || kg::verify_kernel + "
                                    "ordered placement (kg::hit_partition_kernel x 2, kg::group_place_kernel): chunks of whole "
                                    "contigs on three streams" if partitioned else "kg::scan_kernel<false,false,3>"),
                         "kernel_ms": ms_scan,
                         "passes_ms": ({k: float(np.mean(v)) for k, v in pass_ms.items()} if partitioned else None),
                         "strategy": "partitioned" if partitioned else "direct",
                         "alg_bytes_per_residue": b_alg, "slots_per_residue": p_bar, "hits_per_residue": h_bar,
                         "residues_per_launch": int(residues)},
            "stage_ms": {"scan": ms_scan, "order": float(np.mean(order_ms)), "aggregate": float(np.mean(agg_ms)),
                         "device_total": float(np.mean(total_ms))},
        }
        host = None
        if world == 1 and not args.no_cpu_baseline:
            host = host_table(args, rec)
            out["cpu_baseline"] = cpu_baseline(args, rec, seq, off, tab, host)
        if world == 1 and not args.no_extra_configs:
            del seq
            torch.cuda.empty_cache()
            out["extra_configs"] = extra_configs(args, tab, host, dev)
        print(json.dumps(out), flush=True)

    tab.close()
    if multi:
        dist.destroy_process_group()


def host_table(args, rec):
    """The table image (header + records) in host memory: what the CPU oracle reads."""
    import struct
    t0 = time.time()
    host = torch.empty(24 + args.num_sigs * 24, dtype=torch.uint8)
    host[:24] = torch.frombuffer(bytearray(struct.pack("<qqq", args.num_sigs, 24, 1)), dtype=torch.uint8)
    host[24:].view(torch.int32).view(args.num_sigs, 6).copy_(rec)
    log("[bench] cpu_baseline: table copied to host in %.1f s" % (time.time() - t0))
    return host


def extra_configs(args, tab, host, dev):
    """BASELINE configs 2 and 5 behind the headline, same ABI, same timing rule (inputs resident in HBM, CALL / OTU records
    to the host): config 2 = 1000 x 100 kbp uniform DNA against the headline's table; config 5 = 100 Mbp assembled from
    signature 8-mers of <= 32 functions / <= 8 OTUs against its own 20 000 003-slot table (8 M signatures) -- the workload
    that loads processSetOfHits and the OTU votes (KGJ:385-455).  With the CPU baseline enabled, contigs drawn from every
    quarter of each batch are checked against the oracle."""
    from kmergutsjava_amd import hotpath, synth
    steps, warm = max(5, args.steps), 2
    rows = []

    def run(name, tab_x, seq, off, image, note):
        torch.cuda.synchronize()
        for _ in range(warm):
            with tab_x.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
                r.calls(copy=False); r.otu(copy=False)
        sts = []
        t0 = time.perf_counter()
        for _ in range(steps):
            with tab_x.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
                r.calls(copy=False); r.otu(copy=False)
                sts.append(r.stats)
        ms = (time.perf_counter() - t0) / steps * 1e3
        st = sts[-1]
        row = {"workload": name, "note": note, "steps": steps, "ms_per_step": ms, "residues": int(st["residues"]),
               "residues_per_s": st["residues"] / (ms * 1e-3), "hits": int(st["n_hits"]), "hits_per_s": st["n_hits"] / (ms * 1e-3),
               "calls": int(st["n_calls"]), "calls_per_s": st["n_calls"] / (ms * 1e-3),
               "strategy": "partitioned" if st["partitioned"] else "direct", "chunks": int(st["part_chunks"]),
               "device_ms": {"scan": float(np.mean([x["ms_scan"] for x in sts])), "order": float(np.mean([x["ms_order"] for x in sts])),
                             "aggregate": float(np.mean([x["ms_aggregate"] for x in sts])),
                             "total": float(np.mean([x["ms_total"] for x in sts]))},
               "parity_sample": None}
        if image is not None:
            from oracle import kgo
            with tab_x.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
                idx = synth.spread_sample(off, groups=4, per_group=6, max_bp_per_group=700_000)
                sub_off = synth.offsets_of((off[1:] - off[:-1])[idx])
                sub = torch.cat([seq[int(off[i]):int(off[i + 1])] for i in idx]).cpu().numpy()
                o = kgo.run(image, sub, sub_off, lookup_mode=0)
                got = r.subset(idx)
                same = bool(all(got[k].tobytes() == o[k].tobytes() for k in ("hits", "calls", "otu")) and
                            np.array_equal(got["container_hit_start"], o["container_hit_start"]) and
                            np.array_equal(got["container_call_start"], o["container_call_start"]))
            row["parity_sample"] = {"contigs": int(len(idx)), "bp": int(sub_off[-1]), "hits": int(len(o["hits"])),
                                    "calls": int(len(o["calls"])), "identical": same}
            assert same, "%s: GPU records differ from the oracle on the sample contigs" % name
        log("[bench] extra config %s: %.3f ms/step, %d hits, %d CALLs, parity %s" % (name, ms, st["n_hits"], st["n_calls"], row["parity_sample"]))
        rows.append(row)

    seq2, off2 = synth.dna_uniform_config(1000, 100_000, 201, dev)
    run("config 2", tab, seq2, off2, None if host is None else host.numpy(),
        "BASELINE config 2: 1000 x 100 kbp uniform ACGT vs the %d-slot table, -m 5 -g 200" % args.num_sigs)
    del seq2
    seq5, off5, rec5 = synth.high_density_device(1000, 4167, 20_000_003, 8_000_000, 501, True, dev)
    torch.cuda.synchronize()
    with hotpath.SignatureTable.from_device_ptr(rec5.data_ptr(), 20_000_003, dev.index or 0, keepalive=rec5) as tab5:
        run("config 5 (DNA)", tab5, seq5, off5, None if host is None else synth.table_image(rec5),
            "BASELINE config 5: 1000 contigs x 4167 signature 8-mers (100 Mbp, back-translated), <= 32 functions / <= 8 OTUs, "
            "20 000 003-slot table with 8 M signatures, -m 5 -g 200")
    return rows


def cpu_baseline(args, rec, seq, off, tab, host):
    """The C restatement of the reference algorithm (oracle/, literal sorted merge-join, 1 thread) timed on a prefix
    of the same workload against the same table, in batches of <= 20 M query k-mers (the reference's loss-free
    regime, KGJ:108, 832).  Its records are also the checker for the full-size GPU scan: the hit / CALL / OTU records
    of the prefix AND of a second sample of contigs drawn from every chunk of the GPU pipeline must be byte-identical."""
    import struct
    from oracle import kgo
    from kmergutsjava_amd import hotpath, synth
    kgo.build()
    n = int(np.searchsorted(off, args.cpu_sample_bp, side="right"))
    n = max(1, min(n, len(off) - 1))
    sample_off = off[:n + 1].copy()
    sample = seq[:int(sample_off[-1])].cpu().numpy()
    t0 = time.time()
    o = kgo.run(host.numpy(), sample, sample_off, lookup_mode=0)
    wall = time.time() - t0
    phases = o["t_prepare"] + o["t_lookup"] + o["t_group"]
    log("[bench] cpu_baseline: %d residues in %.1f s (prepare %.1f, lookup %.1f, group %.1f)" %
        (o["residues"], wall, o["t_prepare"], o["t_lookup"], o["t_group"]))
    # second row (cpu_baseline.all_cores): what the host can do when it drops the reference's plan -- the same prefix cut into one group of whole
    # contigs per thread, each group through the restatement with direct hash probing instead of the sorted merge-join
    # (ctypes releases the GIL: the groups run in parallel).  Records compared with the single-thread run above.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(args.cpu_threads or min(16, avail), n))          # 16 = a one-GPU box's share of the host's cores
    all_cores = None
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        lens = sample_off[1:] - sample_off[:-1]
        cuts = np.searchsorted(np.cumsum(lens), np.arange(1, threads) * (int(sample_off[-1]) / threads), side="left")
        bounds = np.unique(np.concatenate([[0], cuts, [n]]))

        def part(k):
            a, b = int(bounds[k]), int(bounds[k + 1])
            return kgo.run(host.numpy(), sample[int(sample_off[a]):int(sample_off[b])], sample_off[a:b + 1] - sample_off[a],
                           lookup_mode=1)
        t0 = time.time()
        with ThreadPoolExecutor(len(bounds) - 1) as ex:
            parts = list(ex.map(part, range(len(bounds) - 1)))
        wall_mt = time.time() - t0
        def rebased(kind):
            out = []
            for k, q in enumerate(parts):
                rec = q[kind].copy()
                rec["container"] += 6 * int(bounds[k])
                out.append(rec.tobytes())
            return b"".join(out)
        same_mt = rebased("hits") == o["hits"].tobytes() and rebased("calls") == o["calls"].tobytes()
        all_cores = {"value": o["residues"] / wall_mt, "unit": "residues/s", "cores": len(bounds) - 1, "kind": "port",
                     "seconds": wall_mt, "records_identical_to_single_thread": bool(same_mt),
                     "sample": "the same prefix, one group of whole contigs per thread, direct hash probing (no sort, no merge-join)"}
        log("[bench] cpu_baseline (all cores): %d threads, %.2f s -> %.3g residues/s" % (len(bounds) - 1, wall_mt, all_cores["value"]))
    # parity of the full-size scan (same strategy as the timed steps)
    with tab.scan(None, off, hotpath.Params(), device_ptr=seq.data_ptr()) as r:
        n_chunks = max(1, r.stats["part_chunks"])
        idx = synth.spread_sample(off, groups=max(4, n_chunks), per_group=20, max_bp_per_group=2_500_000)
        sub_off = synth.offsets_of((off[1:] - off[:-1])[idx])
        sub = torch.cat([seq[int(off[i]):int(off[i + 1])] for i in idx]).cpu().numpy()
        o2 = kgo.run(host.numpy(), sub, sub_off, lookup_mode=0)

        def same(got, ora):
            return bool(all(got[k].tobytes() == ora[k].tobytes() for k in ("hits", "calls", "otu")) and
                        np.array_equal(got["container_hit_start"], ora["container_hit_start"]) and
                        np.array_equal(got["container_call_start"], ora["container_call_start"]))
        ok_prefix = same(r.subset(np.arange(n)), o)
        ok_spread = same(r.subset(idx), o2)
        parity = {"prefix_contigs": n, "prefix_hits": int(len(o["hits"])), "prefix_identical": ok_prefix,
                  "chunks": int(r.stats["part_chunks"]), "spread_contigs": int(len(idx)),
                  "spread_contig_range": [int(idx[0]), int(idx[-1])], "spread_hits": int(len(o2["hits"])),
                  "spread_calls": int(len(o2["calls"])), "spread_identical": ok_spread,
                  "identical": ok_prefix and ok_spread,
                  "strategy": "partitioned" if r.stats["partitioned"] else "direct"}
    log("[bench] parity of the full-size scan on the samples: %s" % parity)
    assert parity["identical"], "full-size GPU scan differs from the oracle on the sample contigs"
    return {"value": o["residues"] / phases, "unit": "residues/s", "cores": 1, "kind": "port", "parity_sample": parity,
            "sample": "first %d contigs (%d bp, %d residues, %d query k-mers, in batches of <= 20 M k-mers) of the same "
                      "contig mix against the same table; C restatement of the reference's materialise -> sort by "
                      "(value %% numSigs, value) -> streamed merge-join -> gatherHits, single thread"
                      % (n, int(sample_off[-1]), o["residues"], o["windows_valid"]),
            "seconds": phases, "cpu": _cpu_model(), "host_threads_available": os.cpu_count(), "all_cores": all_cores,
            "phases_s": {"preparation": o["t_prepare"], "lookup": o["t_lookup"], "grouping": o["t_group"]}}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
