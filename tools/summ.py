#!/usr/bin/env python3
"""usage: tools/summ.py gpurun_out/<tag> -- one line per step log that holds tools/one_scan.py JSON lines: ms_scan / ms_total"""
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.log")), key=os.path.getmtime):
    js = [json.loads(l) for l in open(f) if l.startswith("{") and "ms_scan" in l and "ms_total" in l and "n_seqs" in l]
    if js:
        print("%-18s scan %s  total %s" % (os.path.basename(f)[:-4], " ".join("%.2f" % j["ms_scan"] for j in js[1:]),
                                           " ".join("%.2f" % j["ms_total"] for j in js[1:])))
