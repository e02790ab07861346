#!/bin/bash
# Experiment: is the centred-Gram launch limited by HBM re-fetch?  The same kernel with every member reading member 0's
# features (csrc/lab/_build/libhecka_alias.so, -DHECKA_EXP_ALIAS: 16 MB of input instead of 1 GB, all cache-resident).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in 8192 2048; do
  echo "== product library, P=$P"; timeout -k 10 60 python $R/tools/kernel_bench.py gram_center 64 512 $P 30 2>/dev/null || exit 1
  echo "== every member aliased to member 0, P=$P"; HECKA_LIB_PATH=$R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_alias.so timeout -k 10 60 python $R/tools/kernel_bench.py gram_center 64 512 $P 30 2>/dev/null || exit 1
done
for v in product alias; do
  L=""; [ $v = alias ] && export HECKA_LIB_PATH=$R/he-cka-ensembles_amd/csrc/lab/_build/libhecka_alias.so
  timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/alias_$v -o p -- python3 $R/tools/kernel_bench.py gram_center 64 512 8192 3 > /dev/null 2>&1 || exit 1
  echo "== FETCH_SIZE (KiB per launch, raw), $v:"; python3 $R/tools/pmc_summary.py $R/gpurun_out/alias_$v | grep -A1 gram_center
  rm -rf $R/gpurun_out/alias_$v
done
