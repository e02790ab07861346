#!/bin/bash
# Same-box run of tools/tile_ablation.py over the in-tree library and the csrc/lab/_build ablation builds.
R=$GRAFT_REPO_ROOT
timeout -k 10 120 python $R/tools/tile_ablation.py full || exit 1
HECKA_K32=0 timeout -k 10 120 python $R/tools/tile_ablation.py x6tile || exit 1
for f in $R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_*.so; do
  [ -e "$f" ] || continue
  t=$(basename $f .so); t=${t#libhecka_}
  HECKA_LIB_PATH=$f timeout -k 10 120 python $R/tools/tile_ablation.py $t || exit 1
done
