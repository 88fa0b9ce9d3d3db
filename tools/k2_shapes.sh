#!/bin/bash
# developer tool: K2 tile / halo shapes (devel library, CLOOPS_K2_SHAPE) at the sparse and the dense workloads
export CLOOPS_DEVEL_LIB=1
for sh in 0 3 4 6; do
  echo "== 5M eps 2000 minPts 5 shape $sh"; CLOOPS_K2_SHAPE=$sh python tools/quick_timing.py 5e6 2000 5 v2 2>&1 | grep -E "iter 3|K2:"
done
for cfg in "5000 50 0" "7500 30 5000" "10000 20 5000"; do
  for sh in 0 1 2 4 5 7; do
    echo "== dense $cfg shape $sh"; CLOOPS_K2_SHAPE=$sh python tools/dense_run.py $cfg 4 2>&1 | grep -E "iter 3|K2:"
  done
done
