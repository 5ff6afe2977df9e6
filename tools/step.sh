#!/bin/bash
# Helper for the GPU calls: run the given steps in order, each under its own timeout; stop at the first step
# that was killed by its timeout (no GPU step is started after a hang).  usage: r03_step.sh <tag> ; steps read from stdin
# as lines "name|timeout_s|command".
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
while IFS='|' read -r name tmo cmd; do
    [ -z "$name" ] && continue
    echo "== $name (timeout $tmo s): $cmd"
    t0=$(date +%s)
    ( cd $ROOT && timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2> $OUT/$name.err )
    rc=$?
    echo "   rc=$rc in $(( $(date +%s) - t0 )) s"
    tail -n 3 $OUT/$name.log | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   TIMEOUT: stopping here"; tail -n 5 $OUT/$name.err; exit 1; fi
    if [ $rc -ne 0 ]; then tail -n 12 $OUT/$name.err | cut -c1-300; fi
done
exit 0
