set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
export VTC_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03b_stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-extra --no-cpu --no-sweep > $O/r03b_stats.json 2> $O/r03b_stats.err || exit 1
unset VTC_OVERLAP
rocprofv3 --kernel-trace --output-format csv -d $O/r03b_step -- python3 $R/tools/step_kernels.py 64 4 > $O/r03b_step.log 2>&1 || exit 1
cd $R
find $O/r03b_stats -name "*kernel_trace.csv" -size +20M -delete
python bench.py > $O/r03b_bench.json 2> $O/r03b_bench.err || exit 1
echo profiles done
