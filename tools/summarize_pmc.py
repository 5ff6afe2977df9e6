#!/usr/bin/env python3
"""Turn the PMC passes of tools/collect_profiles.sh into profiles/traffic.json + per-kernel tables.
usage: summarize_pmc.py gpurun_out/<tag> <round>"""
import collections, csv, glob, json, os, shutil, sys
R, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def agg(path):
    rows = list(csv.DictReader(open(glob.glob(path)[0])))
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        a[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a

f, w = agg(R + "/pmc_fetch/runc/*_counter_collection.csv"), agg(R + "/pmc_write/runc/*_counter_collection.csv")
q, h = agg(R + "/pmc_rdreq/runc/*_counter_collection.csv"), agg(R + "/pmc_hit/runc/*_counter_collection.csv")
bench = json.load(open(R + "/pmc_fetch.json"))
strategy = bench["roofline"]["strategy"]
scan_kernels = [k for k in f if any(x in k for x in (("part_scatter", "bucket_tag", "bucket_index", "sub_scatter", "sub_probe", "sub_index", "verify_kernel", "overflow_probe", "row_geo", "hit_hist", "group_scan", "hit_partition", "group_place", "lowc_blocks", "chunk_base") if strategy == "partitioned" else ("scan_kernel<false, false",)))]
per_kernel = {}
tot_fetch = tot_write = 0.0
steps = 1      # the PMC runs use --steps 1 --warmup 0 plus the counters launch: take the launches of the LAST scan
for k in sorted(f):
    if not k.startswith(("void kg::", "kg::")):
        continue
    n = len(f[k]["FETCH_SIZE"])
    per_kernel[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": sum(f[k]["FETCH_SIZE"]) / n,
                     "WRITE_SIZE_KB_per_launch": sum(w[k]["WRITE_SIZE"]) / max(1, len(w[k]["WRITE_SIZE"])),
                     "TCC_EA0_RDREQ_per_launch": sum(q[k]["TCC_EA0_RDREQ_sum"]) / max(1, len(q[k]["TCC_EA0_RDREQ_sum"])),
                     "TCC_HIT_per_launch": sum(h[k]["TCC_HIT_sum"]) / max(1, len(h[k]["TCC_HIT_sum"])),
                     "TCC_MISS_per_launch": sum(h[k]["TCC_MISS_sum"]) / max(1, len(h[k]["TCC_MISS_sum"]))}
# one scan = all launches of the scan-stage kernels in one bench step; the PMC bench run makes 2 scans (counters + 1 timed)
scans = 2
# (round 4) the counters scan probes the tags (bucket_tag_kernel<true> + the <.., true> verify kernels), the timed scan the byte
# home index (bucket_index_kernel + the <.., false> verify kernels): the timed scan = its own kernels whole + half of the
# launches of the kernels both scans share
COUNTERS_ONLY = ("bucket_tag_kernel<true>", "verify_kernel<false, true>", "verify_kernel<true, true>", "overflow_probe_kernel<false, true>",
                 "overflow_probe_kernel<true, true>")
TIMED_ONLY = ("bucket_index_kernel", "bucket_tag_kernel<false>", "verify_kernel<false, false>", "verify_kernel<true, false>",
              "overflow_probe_kernel<false, false>", "overflow_probe_kernel<true, false>")
for k in scan_kernels:
    if "true>" in k and strategy == "direct":
        continue
    if any(x in k for x in COUNTERS_ONLY):
        continue
    share = 1 if any(x in k for x in TIMED_ONLY) else scans
    tot_fetch += sum(f[k]["FETCH_SIZE"]) / share * (2 if (strategy == "direct" and len(f[k]["FETCH_SIZE"]) == 1) else 1)
    tot_write += sum(w[k]["WRITE_SIZE"]) / share * (2 if (strategy == "direct" and len(w[k]["WRITE_SIZE"]) == 1) else 1)
out = {"round": rnd, "total_bp": bench["config"].get("total_bp_rank0", bench["config"].get("total_bp_per_gpu")),
       "num_sigs": bench["config"]["num_sigs"], "strategy": strategy,
       "scan_stage_kernels": scan_kernels, "FETCH_SIZE_KB_per_scan": tot_fetch, "WRITE_SIZE_KB_per_scan": tot_write,
       "hbm_bytes_per_launch": (2 * tot_fetch + tot_write) * 1024,
       "correction": "gfx950: FETCH_SIZE tallies every TCC_EA0_RDREQ at 64 B but the requests are 128-B lines (MI355X_MICROARCH.md 'HBM': "
                     "double FETCH_SIZE; confirmed in this access pattern by profiles/r01 calibration: aligned random 16-B loads make 1.000 "
                     "request per load, unaligned 1.114 = 1 + 15/128).  WRITE_SIZE is exact.  'per launch' = per scan stage (one bench step).",
       "per_kernel": per_kernel,
       "collected_with": "rocprofv3 --pmc <group> --kernel-include-regex kg:: -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                         "(separate passes: FETCH_SIZE; WRITE_SIZE; TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; TCC_HIT_sum TCC_MISS_sum)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(glob.glob(R + "/trace/runc/*_kernel_stats.csv")[0], os.path.join(ROOT, "profiles", rnd + "_kernel_stats.csv"))
shutil.copy(R + "/bench.json", os.path.join(ROOT, "profiles", rnd + "_bench.json"))
shutil.copy(R + "/bench_traced.json", os.path.join(ROOT, "profiles", rnd + "_bench_traced.json"))
for n in ("fetch", "write", "rdreq", "hit"):
    shutil.copy(glob.glob(R + "/pmc_%s/runc/*_counter_collection.csv" % n)[0], os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (rnd, n)))
print(json.dumps({k: out[k] for k in ("strategy", "FETCH_SIZE_KB_per_scan", "WRITE_SIZE_KB_per_scan", "hbm_bytes_per_launch")}, indent=1))
