#!/bin/bash
# round-3 rocprofv3 evidence, run on the GPU box from the repo root (gpurun): stats + PMC passes of the bench command, sweep
# stats at 10k / 50k, a steady-state kernel trace.  Outputs under gpurun_out/r03_*; summaries are made here by tools/*.py.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-extra --no-cpu --no-sweep > $O/r03_stats.json 2> $O/r03_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep > /dev/null 2> $O/r03_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep > /dev/null 2> $O/r03_write.err || exit 1
unset VTC_OVERLAP
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_sweep10k -- python3 $R/tools/sweep_profile.py 10000 30 > $O/r03_sweep10k.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_sweep50k -- python3 $R/tools/sweep_profile.py 50000 10 > $O/r03_sweep50k.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/r03_step -- python3 $R/tools/step_kernels.py 64 4 > $O/r03_step.log 2>&1 || exit 1
cd $R
# keep only the small CSVs (the merge back is capped)
find $O/r03_stats $O/r03_sweep10k $O/r03_sweep50k -name "*kernel_trace.csv" -size +20M -delete
echo profiles done
