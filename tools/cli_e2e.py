#!/usr/bin/env python3
"""End-to-end drop-in path, files in -> report out, through the native front end (kmer_guts):
FASTA (80-column lines) on disk -> parse -> H2D -> scan -> CALL/OTU text.  PCIe- and IO-inclusive; never the
bench `value`.  1 Gbp of contigs (10 000 x 100 kbp) against a 100 000 007-slot table file (2.4 GB)."""
import json, os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmergutsjava_amd import synth, build
work = os.environ.get("E2E_DIR", "/tmp/kg_e2e"); os.makedirs(work, exist_ok=True)
n_contigs, L = int(os.environ.get("E2E_CONTIGS", "10000")), 100000
dev = torch.device("cuda", 0)
rec, placed, keys = synth.random_table(100_000_007, 0.5, 202, dev); del keys
synth.write_data_dir(os.path.join(work, "d"), synth.table_image(rec), 1000); del rec
seq, off = synth.dna_uniform_config(n_contigs, L, 201, dev)
rows = seq.cpu().numpy().reshape(n_contigs, L // 80, 80)
nl = np.full((n_contigs, L // 80, 1), 10, dtype=np.uint8)
body = np.concatenate([rows, nl], axis=2).reshape(n_contigs, -1)
fa = os.path.join(work, "q.fna")
with open(fa, "wb") as f:
    for k in range(n_contigs):
        f.write(b">contig%d synthetic\n" % k); f.write(body[k].tobytes())
del seq, rows, body
cli = build.build_cli()
out = {"fasta_bytes": os.path.getsize(fa), "bp": n_contigs * L}
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([cli, "-D", os.path.join(work, "d"), "-q", fa, "-o", os.path.join(work, "out.txt")], stdout=subprocess.PIPE, check=True)
    out["run%d" % rep] = {"wall_s": time.perf_counter() - t0, "info": r.stdout.decode().strip().split("\n")}
out["report_bytes"] = os.path.getsize(os.path.join(work, "out.txt"))
out["residues_per_s_end_to_end"] = 2 * n_contigs * L / out["run1"]["wall_s"]
print(json.dumps(out, indent=1))
