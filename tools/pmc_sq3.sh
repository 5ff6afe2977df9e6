#!/bin/bash
# SQ counters of the scan passes, one chunk (KG_PART_CHUNKS=1) so that the kernels do not overlap; one full-size scan
# (tools/one_scan.py) per pass.  Usage (GPU box, repo root): bash tools/pmc_sq3.sh <tag> [ENV=VAL ...]
set -u
TAG=${1:-sq}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp SW_REPS=1
export KG_PART_CHUNKS=${KG_PART_CHUNKS:-1}
for kv in "$@"; do export "$kv"; done
cd /tmp
pmc() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-include-regex "kg::" --output-format csv -d $OUT/pmc_$name -o c -- \
        python3 $ROOT/tools/one_scan.py > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || { echo "pmc $name rc=$?"; return 1; }
}
pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH &&
pmc cycles SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_BUSY_CYCLES
python3 - $OUT <<'PY'
import collections, csv, glob, json, sys
R = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(R + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kg::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
keep = {k: v for k, v in tot.items() if v.get("SQ_WAVE_CYCLES", 0) > 1e8}
json.dump(keep, open(R + "/summary.json", "w"), indent=1)
for k, v in keep.items():
    print(k, {c: "%.3g" % x for c, x in sorted(v.items())})
PY
find $OUT -type f ! -name summary.json ! -name "*.err" -delete 2>/dev/null
