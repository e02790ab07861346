#!/bin/bash
# Same-box A/B of the small HBM-side stages (tools/small_ab.py): in-tree library and every csrc/lab/_build/libhecka_*.so, two rounds.
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  timeout -k 10 120 python $R/tools/small_ab.py tree || exit 1
  HECKA_COMBINE_X3=0 timeout -k 10 120 python $R/tools/small_ab.py tree-f32 || exit 1
  for f in $R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_*.so; do
    [ -e "$f" ] || continue
    t=$(basename $f .so); t=${t#libhecka_}
    HECKA_LIB_PATH=$f timeout -k 10 120 python $R/tools/small_ab.py $t || exit 1
  done
done
