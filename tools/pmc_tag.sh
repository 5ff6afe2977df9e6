#!/bin/bash
# Memory-side counters of the partitioned passes, one chunk (kernels do not overlap).  Usage: bash tools/pmc_tag.sh <tag>
set -u
TAG=${1:-tagpmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export KG_PART_CHUNKS=1
cd /tmp
pmc() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-include-regex "bucket_tag|part_scatter|verify_kernel|place_unordered" --output-format csv -d $OUT/pmc_$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc $name rc=$?"
}
pmc tcc1 TCC_REQ_sum TCC_READ_sum TCC_READ_SECTORS_sum &&
pmc tcc2 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum &&
pmc ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | head
