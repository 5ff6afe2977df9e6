#!/bin/bash
# Runs on the GPU box (from the repo root): the default bench, its rocprofv3 kernel-trace summary, and the
# PMC passes (one counter group per pass, as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes).
# Everything lands under gpurun_out/<tag>/; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp

echo "== bench (default flags)"
timeout -k 10 600 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench rc=$?"
cat $OUT/bench.json | cut -c1-300

echo "== rocprofv3 --kernel-trace --stats (same command)"
# (--no-extra-configs: configs 2 and 5 launch the same kernels on smaller batches and would drag their averages down)
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-extra-configs \
    > $OUT/bench_traced.json 2> $OUT/bench_traced.err || echo "trace rc=$?"

echo "== counters available"
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_HIT|TCC_MISS|TCC_REQ" | head -40 > $OUT/counters_list.txt

pmc() {   # name, counters...
    local name=$1; shift
    timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-include-regex "kg::" --output-format csv -d $OUT/pmc_$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc $name rc=$?"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc hit TCC_HIT_sum TCC_MISS_sum

echo "== calibration: known number of random 16-byte reads (tools/gather_ceiling.bin 1400 MiB, aligned and not)"
for al in 1 0; do
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch_al$al -- $ROOT/tools/gather_ceiling.bin 1400 3 $al 64 8192 \
      > $OUT/cal_fetch_al$al.json 2>/dev/null || echo "cal rc=$?"
  timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/cal_rdreq_al$al -- $ROOT/tools/gather_ceiling.bin 1400 3 $al 64 8192 \
      > $OUT/cal_rdreq_al$al.json 2>/dev/null || echo "cal rc=$?"
done
find $OUT -name "*.csv" | head -50
du -sh $OUT
