#!/bin/bash
# PMC passes over the partitioned path's kernels (tuning aid).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_part3; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export KG_PARTITION=1 SW_REPS=1
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "bucket_tag|part_scatter|verify_kernel|place_unordered" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/one_scan.py > /dev/null 2> $OUT/$name.err || echo "rc=$? for $grp"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-50s %-26s %.6g" % (k[0], k[1], sum(v) / len(v)))
PY
