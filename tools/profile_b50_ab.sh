#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/b50_new $O/b50_base
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b50_new -- python3 $R/tools/step_time.py 50 > $O/b50_new.log 2>&1 || echo "(non-zero exit)"
export VTC_HIP_LIB=$R/vtc_amd/lib/variants/libvtc_base.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b50_base -- python3 $R/tools/step_time.py 50 > $O/b50_base.log 2>&1 || echo "(non-zero exit)"
cd $R
for d in b50_new b50_base; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d"; head -16 $f | cut -d, -f1-4 | cut -c1-150
  find $O/$d -name "*kernel_trace.csv" -delete
done
