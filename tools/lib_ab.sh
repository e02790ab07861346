#!/bin/bash
# Same-box A/B of the in-tree libhecka.so against the builds under csrc/lab/_build/ (libhecka_<tag>.so): median times of the
# headline workload's two GEMM launches (tools/tile_ablation.py), three alternating rounds.
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  timeout -k 10 120 python $R/tools/tile_ablation.py tree || exit 1
  for f in $R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_*.so; do
    [ -e "$f" ] || continue
    t=$(basename $f .so); t=${t#libhecka_}
    HECKA_LIB_PATH=$f timeout -k 10 120 python $R/tools/tile_ablation.py $t || exit 1
  done
done
