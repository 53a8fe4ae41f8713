#!/bin/bash
# one PMC pass of the bench command for per-kernel matrix-pipe occupancy (round-5 final kernels)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r05_mfma_step
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r05_mfma_step -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r05_mfma_step.err || echo "(non-zero exit)"
cd $R
python3 tools/pmc_mfma_busy.py $O/r05_mfma_step > $O/r05_mfma_step.md
find $O/r05_mfma_step -name "*.csv" -size +20M -delete
echo done
