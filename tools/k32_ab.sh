#!/bin/bash
# Same-box A/B of the 16x16x32 tile (HECKA_K32=1, default) against the 32x32x16 one: per-kernel times from rocprofv3.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
FAST="--no-cpu-baseline --no-extras --no-parity"
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export HECKA_K32=$v
  rm -rf $O/k32ab_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k32ab_$v -o bench -- python3 $R/bench.py --steps 10 --warmup 3 $FAST > $O/k32ab_$v.json 2> $O/k32ab_$v.err || exit 1
  find $O/k32ab_$v -name "*kernel_stats.csv" -exec cp {} $O/k32ab_${v}_stats.csv \;
  rm -rf $O/k32ab_$v
  echo "HECKA_K32=$v"; python $R/tools/top_kernels.py $O/k32ab_${v}_stats.csv 6
done
