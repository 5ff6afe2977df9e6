#!/bin/bash
# SQ counters of the partitioned passes, one chunk (KG_PART_CHUNKS=1) so that the kernels do not overlap.
# Usage (GPU box, repo root): bash tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export KG_PART_CHUNKS=${KG_PART_CHUNKS:-1}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_list.txt
pmc() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-include-regex "kg::" --output-format csv -d $OUT/pmc_$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc $name rc=$?"
}
pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH &&
pmc cycles SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/traced.json 2> $OUT/traced.err
find $OUT -name "*.csv" | head -20
