#!/bin/bash
# A/B of the tag-pass variants (tuning aid): scan stage of the 1 Gbp workload, then the parity tests
out=gpurun_out/r02/ab; mkdir -p $out
export SW_REPS=4
for v in 1 0; do
  KG_TAG_PIPE=$v timeout -k 10 200 python3 tools/one_scan.py > $out/pipe$v.log 2>&1 || { tail -5 $out/pipe$v.log; exit 1; }
  grep -o '"ms_scan": [0-9.]*' $out/pipe$v.log | tr '\n' ' '; echo " <- KG_TAG_PIPE=$v"
done
KG_TAG_PIPE=1 KG_PROBE_GRID=2048 timeout -k 10 200 python3 tools/one_scan.py > $out/pipe1_g2048.log 2>&1; grep -o '"ms_scan": [0-9.]*' $out/pipe1_g2048.log | tr '\n' ' '; echo " <- pipe, grid 2048"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "fullsize or parity or fuzz" > $out/tests.log 2>&1; tail -3 $out/tests.log
