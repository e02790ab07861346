#!/bin/bash
# One GPU call: short bench (parity on, no extras / CPU baseline) + rocprofv3 kernel stats.  bash tools/gpu_quick.sh <tag>
set -o pipefail
TAG=${1:-quick}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-extras > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
rm -rf $O/${TAG}_prof
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity > /dev/null 2>> $O/${TAG}_bench.err || exit 1
cd $R
python tools/prof_summary.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_summary.txt
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench.json'))
print('evals/s %.1f  median %.3f ms  dX %.3f ms frac %.3f  parity loss %.1e grad %.1e' % (d['value'], d['median_ms_device'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity']['loss_rel'], d['parity']['grad_rel_max']))
PY
head -6 gpurun_out/${TAG}_summary.txt
