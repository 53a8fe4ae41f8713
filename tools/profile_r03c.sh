#!/bin/bash
# PMC passes (HBM traffic) of the final round-3 bench command: separate --pmc runs, no trace domains (gpurun rule)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03c_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep > /dev/null 2> $O/r03c_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03c_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep > /dev/null 2> $O/r03c_write.err || exit 1
echo pmc done
