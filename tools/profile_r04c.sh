#!/bin/bash
# measured peaks of the box (incl. the sustained MFMA run) + one PMC pass of the bench command for per-kernel matrix-pipe occupancy
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r04_mfma_step
./tools/probes/peaks > $O/r04_peaks.txt 2>&1 || echo "(peaks non-zero exit)"
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r04_mfma_step -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r04_mfma_step.err || echo "(non-zero exit)"
cd $R
python3 tools/pmc_mfma_busy.py $O/r04_mfma_step > $O/r04_mfma_step.md
find $O/r04_mfma_step -name "*.csv" -size +20M -delete
echo done
