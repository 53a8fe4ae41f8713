#!/bin/bash
# the three bench passes of tools/profile_r04.sh alone (stats, FETCH_SIZE, WRITE_SIZE)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r04_stats $O/r04_fetch $O/r04_write
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-extra --no-cpu --no-sweep --no-independence > $O/r04_stats.json 2> $O/r04_stats.err || echo "(non-zero exit)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r04_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r04_fetch.err || echo "(non-zero exit)"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r04_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r04_write.err || echo "(non-zero exit)"
cd $R
find $O/r04_stats -name "*kernel_trace.csv" -size +20M -delete
echo profiles done
