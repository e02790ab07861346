#!/bin/bash
# Same-box A/B of one rank's compute at 8- and 4-way member shards (tools/shard_emulate.py, no stand-in copies): in-tree
# library against every csrc/lab/_build/libhecka_*.so, three alternating rounds.
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  echo "tree:"; timeout -k 10 200 python $R/tools/shard_emulate.py --modes slice --worlds 4,8 --no-exchange-copies 2>/dev/null || exit 1
  for f in $R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_*.so; do
    [ -e "$f" ] || continue
    echo "$(basename $f):"; HECKA_LIB_PATH=$f timeout -k 10 200 python $R/tools/shard_emulate.py --modes slice --worlds 4,8 --no-exchange-copies 2>/dev/null || exit 1
  done
done
