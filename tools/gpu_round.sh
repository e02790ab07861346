#!/bin/bash
# One GPU call: bench, rocprofv3 kernel stats, HBM-traffic and MFMA-busy PMC passes for bench.py.
# Usage on the GPU box (via gpurun):  bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
FAST="--no-cpu-baseline --no-extras --no-parity"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
rm -rf $O/${TAG}_prof $O/${TAG}_pmcR $O/${TAG}_pmcW $O/${TAG}_pmcM
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 $FAST > $O/${TAG}_bench_profiled.json 2>> $O/${TAG}_bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmcR -o p -- python3 $R/bench.py --steps 3 --warmup 1 $FAST > /dev/null 2>> $O/${TAG}_bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmcW -o p -- python3 $R/bench.py --steps 3 --warmup 1 $FAST > /dev/null 2>> $O/${TAG}_bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmcM -o p -- python3 $R/bench.py --steps 3 --warmup 1 $FAST > /dev/null 2>> $O/${TAG}_bench.err || exit 1
cd $R
python tools/prof_summary.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_summary.txt
python tools/pmc_summary.py gpurun_out/${TAG}_pmcR > gpurun_out/${TAG}_pmc_fetch.txt
python tools/pmc_summary.py gpurun_out/${TAG}_pmcW > gpurun_out/${TAG}_pmc_write.txt
python tools/pmc_summary.py gpurun_out/${TAG}_pmcM > gpurun_out/${TAG}_pmc_mfma.txt
cp $O/${TAG}_prof/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null || find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf $O/${TAG}_pmcR $O/${TAG}_pmcW $O/${TAG}_pmcM
cut -c1-600 gpurun_out/${TAG}_bench.json; head -30 gpurun_out/${TAG}_summary.txt; cat gpurun_out/${TAG}_pmc_fetch.txt gpurun_out/${TAG}_pmc_write.txt gpurun_out/${TAG}_pmc_mfma.txt
