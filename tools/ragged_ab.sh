#!/bin/bash
# A/B of the ragged shapes: split-precision tile (HECKA_RAGGED_X3=1, default) against the fp32-MFMA kernels (=0).
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "5 300 864" "5 300 256" "30 100 40" "128 400 256" "128 400 864" "128 300 864" "10 300 864"; do
  for k in gram_center gram_bwd; do
    for v in 0 1; do
      echo -n "RAGGED_X3=$v  "
      HECKA_RAGGED_X3=$v timeout -k 10 60 python $R/tools/kernel_bench.py $k $shape 50 2>/dev/null || exit 1
    done
  done
done
