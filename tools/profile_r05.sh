#!/bin/bash
# round-5 rocprofv3 evidence, run on the GPU box from the repo root (gpurun): kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of the
# bench command (config 3, towers serialised), the same stats for the fp32 mode, and stats + both PMC passes of the N x N sweep at 10k / 50k.
# Counter passes carry --pmc only (no trace domains): MI355X_MICROARCH.md "HBM" recipe; FETCH_SIZE x 2 on gfx950.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r05_stats $O/r05_fetch $O/r05_write $O/r05_f32_stats $O/r05_sw10_stats $O/r05_sw50_stats $O/r05_sw10_fetch $O/r05_sw10_write $O/r05_sw50_fetch $O/r05_sw50_write
cd /tmp
export VTC_OVERLAP=0
B="--no-extra --no-cpu --no-sweep --no-independence"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_stats -- python3 $R/bench.py --steps 8 --warmup 2 $B > $O/r05_stats.json 2> $O/r05_stats.err || echo "(stats: non-zero exit)"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r05_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2> $O/r05_fetch.err || echo "(fetch: non-zero exit)"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r05_write -- python3 $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2> $O/r05_write.err || echo "(write: non-zero exit)"
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_f32_stats -- python3 $R/bench.py --dtype f32 --batch 256 --steps 3 --warmup 1 $B > $O/r05_f32_stats.json 2> $O/r05_f32_stats.err || echo "(f32 stats: non-zero exit)"
echo "f32 done"
for N in 10 50; do
  REPS=30; [ $N = 50 ] && REPS=8
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_sw${N}_stats -- python3 $R/tools/sweep_profile.py ${N}000 $REPS > $O/r05_sw${N}.log 2>&1 || echo "(sweep $N stats: non-zero exit)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r05_sw${N}_fetch -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N fetch: non-zero exit)"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r05_sw${N}_write -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N write: non-zero exit)"
done
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
echo profiles done
