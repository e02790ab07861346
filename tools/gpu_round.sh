#!/bin/bash
# One GPU call: bench (default / serial layer streams), rocprofv3 kernel stats, HBM-traffic PMC passes for bench.py.
# Usage on the GPU box (via gpurun):  bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
HECKA_LAYER_STREAMS=1 timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_serial.json 2>> $O/${TAG}_bench.err || exit 1
rm -rf $O/${TAG}_prof $O/${TAG}_pmcR $O/${TAG}_pmcW
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_profiled.json 2>> $O/${TAG}_bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmcR -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/${TAG}_bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmcW -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/${TAG}_bench.err || exit 1
cd $R
python tools/prof_summary.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_summary.txt
python tools/pmc_summary.py gpurun_out/${TAG}_pmcR > gpurun_out/${TAG}_pmc_fetch.txt
python tools/pmc_summary.py gpurun_out/${TAG}_pmcW > gpurun_out/${TAG}_pmc_write.txt
rm -rf $O/${TAG}_pmcR $O/${TAG}_pmcW
cat gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench_serial.json gpurun_out/${TAG}_summary.txt gpurun_out/${TAG}_pmc_fetch.txt gpurun_out/${TAG}_pmc_write.txt
