#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
rm -rf $O/r05_hardprof2 $O/r05_hardprof4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_hardprof2 -- python3 $R/tools/sweep_time_hard.py 10000 9.0 > $O/r05_hardprof2.log 2>&1
export VTC_SWEEP_PLANES=4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_hardprof4 -- python3 $R/tools/sweep_time_hard.py 10000 9.0 > $O/r05_hardprof4.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
echo done
