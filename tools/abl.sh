#!/bin/bash
# tag-pass ablation (tuning aid): per-variant kernel durations, passes serialised (1 chunk) and pipelined (4 chunks).
# Variant libraries first, here (they travel with the snapshot):
#   git apply tools/tag_ablate.patch
#   for k in 1 3 5 9 13; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKG_ABL=$k -o tools/abl/libkg_abl$k.so \
#       kmergutsjava_amd/csrc/kmerguts_hip.hip -lz -lpthread; done; git apply -R tools/tag_ablate.patch
# KG_ABL bits: 1 no candidate output (so no verification / placement either), 2 no tag compare, 4 synthetic entries (no
# entry stream from HBM), 8 no tag loads.  The results of a variant are wrong by construction; only times are read.
out=gpurun_out/r02/abl; mkdir -p $out
export TMPDIR=/tmp SW_REPS=3
for k in 0 1 3 5 9 13; do
  if [ $k = 0 ]; then unset KG_LIB_PATH; else export KG_LIB_PATH=$PWD/tools/abl/libkg_abl$k.so; fi
  KG_PART_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$k -o t -- python3 tools/one_scan.py > $out/one_$k.log 2>&1 || { echo "variant $k failed"; tail -5 $out/one_$k.log; exit 1; }
  find $out/p$k -type f ! -name "*kernel_stats.csv" -delete
  timeout -k 10 200 python3 tools/one_scan.py > $out/four_$k.log 2>&1 || exit 1
  echo "variant $k done"
done
