#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r04_b50
cd /tmp
export VTC_OVERLAP=${VTC_OVERLAP:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_b50 -- python3 $R/tools/step_time.py 50 > $O/r04_b50.log 2>&1 || echo "(non-zero exit)"
cd $R
f=$(find $O/r04_b50 -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-200
find $O/r04_b50 -name "*kernel_trace.csv" -size +30M -delete
