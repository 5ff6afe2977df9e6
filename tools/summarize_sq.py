#!/usr/bin/env python3
"""SQ counter passes of tools/pmc_sq.sh -> profiles/<round>_sq_summary.json: wave-instructions per kind and the share of
the stage's SIMD time that VALU issue takes.  usage: summarize_sq.py gpurun_out/<tag> <round> [scans]"""
import collections, csv, glob, json, os, sys
R, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(glob.glob(path)[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_BUSY_CYCLES"):
            n[k] += 1
    return a, n
ins, n1 = agg(R + "/pmc_insts/runc/*_counter_collection.csv")
cyc, n2 = agg(R + "/pmc_cycles/runc/*_counter_collection.csv")
out = {"round": rnd, "note": "rocprofv3 --pmc, bench.py --steps 1 --warmup 0 --no-cpu-baseline: TWO scans of the 1 Gbp workload (the "
       "counters scan with KG_F_COUNTERS kernels and one timed scan); kernels without a COUNTERS variant appear with both scans' "
       "launches.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles (MI355X_MICROARCH.md): one wave64 VALU instruction "
       "holds its SIMD for one quad-cycle = 4 clocks.", "kernels": {}}
for k in sorted(ins):
    if "kg::" not in k:
        continue
    out["kernels"][k] = {"launches": n1[k], **{c: v for c, v in ins[k].items()}, **{c: v for c, v in cyc.get(k, {}).items()}}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_sq_counters.json"), "w"), indent=1)
tot = collections.defaultdict(float)
for k, v in out["kernels"].items():
    for c, x in v.items():
        if c != "launches":
            tot[c] += x
print(json.dumps({c: "%.4g" % x for c, x in tot.items()}, indent=1))
