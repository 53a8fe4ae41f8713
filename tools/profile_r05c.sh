#!/bin/bash
# round-5 sweep evidence after the recall-rank path: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of tools/sweep_profile.py at 10k / 50k
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
for N in 10 50; do
  REPS=30; [ $N = 50 ] && REPS=8
  rm -rf $O/r05c_sw${N}_stats $O/r05c_sw${N}_fetch $O/r05c_sw${N}_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05c_sw${N}_stats -- python3 $R/tools/sweep_profile.py ${N}000 $REPS > $O/r05c_sw${N}.log 2>&1 || echo "(sweep $N stats: non-zero exit)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r05c_sw${N}_fetch -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N fetch: non-zero exit)"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r05c_sw${N}_write -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N write: non-zero exit)"
done
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
echo profiles done
