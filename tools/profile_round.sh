#!/bin/bash
# rocprofv3 evidence of one round, run on the GPU box from the repo root (gpurun):
#   tools/profile_round.sh TAG COMMIT [parts...]      parts (default: all): stats pmc mfma f32 sweep
#   stats  kernel-trace stats of the bench command (config 3, towers serialised)            -> gpurun_out/TAG_stats
#   pmc    FETCH_SIZE and WRITE_SIZE passes of the same command (separate passes)            -> gpurun_out/TAG_fetch, TAG_write
#   mfma   SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE pass of the same command                -> gpurun_out/TAG_mfma_step(.md)
#   f32    kernel-trace stats of the fp32 mode at B = 256                                     -> gpurun_out/TAG_f32_stats
#   sweep  kernel-trace stats + both PMC passes of tools/sweep_profile.py at 10k / 50k        -> gpurun_out/TAG_sw{10,50}_{stats,fetch,write}
# and the summaries the repo commits under profiles/ (tools/summarize_*.py): profiles/TAG_config3{.md,_traffic.json,_kernel_stats.csv},
# TAG_f32_config3.md, TAG_sweep_{10k,50k}.md, TAG_sweep_traffic.{json,md}.
# Counter passes carry --pmc only (no trace domains) and put python3 itself behind `--`: MI355X_MICROARCH.md "HBM" recipe and the pool's
# rules; FETCH_SIZE x 2 on gfx950.  (Replaces the per-round profile_r03*.sh ... profile_r05*.sh.)
set -o pipefail
export TMPDIR=/tmp
TAG=$1; COMMIT=$2; shift 2
PARTS="${*:-stats pmc mfma f32 sweep}"
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export VTC_OVERLAP=0
B="--no-extra --no-cpu --no-sweep --no-independence"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has stats; then
  rm -rf $O/${TAG}_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --steps 8 --warmup 2 $B > $O/${TAG}_stats.json 2> $O/${TAG}_stats.err || echo "(stats: non-zero exit)"
  echo "stats done"
fi
if has pmc; then
  rm -rf $O/${TAG}_fetch $O/${TAG}_write
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2> $O/${TAG}_fetch.err || echo "(fetch: non-zero exit)"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2> $O/${TAG}_write.err || echo "(write: non-zero exit)"
  echo "pmc done"
fi
if has mfma; then
  rm -rf $O/${TAG}_mfma_step
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_mfma_step -- python3 $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2> $O/${TAG}_mfma_step.err || echo "(mfma: non-zero exit)"
  (cd $R && python3 tools/pmc_mfma_busy.py $O/${TAG}_mfma_step > $O/${TAG}_mfma_busy_step.md)
  echo "mfma done"
fi
if has f32; then
  rm -rf $O/${TAG}_f32_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_f32_stats -- python3 $R/bench.py --dtype f32 --batch 256 --steps 3 --warmup 1 $B > $O/${TAG}_f32_stats.json 2> $O/${TAG}_f32_stats.err || echo "(f32 stats: non-zero exit)"
  echo "f32 done"
fi
if has sweep; then
  for N in 10 50; do
    REPS=30; [ $N = 50 ] && REPS=8
    rm -rf $O/${TAG}_sw${N}_stats $O/${TAG}_sw${N}_fetch $O/${TAG}_sw${N}_write
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_sw${N}_stats -- python3 $R/tools/sweep_profile.py ${N}000 $REPS > $O/${TAG}_sw${N}.log 2>&1 || echo "(sweep $N stats: non-zero exit)"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_sw${N}_fetch -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N fetch: non-zero exit)"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_sw${N}_write -- python3 $R/tools/sweep_profile.py ${N}000 4 > /dev/null 2>&1 || echo "(sweep $N write: non-zero exit)"
  done
  echo "sweep done"
fi
cd $R
mkdir -p $O/summ
if has stats && has pmc; then
  python3 tools/summarize_profile.py $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_write $O/summ/${TAG}_config3 13 config3 $COMMIT || echo "(summary config3 failed)"
fi
if has f32; then
  python3 tools/summarize_profile.py $O/${TAG}_f32_stats /nonexistent /nonexistent $O/summ/${TAG}_f32_config3 7 config3_f32 $COMMIT || echo "(summary f32 failed)"
fi
if has sweep; then
  python3 tools/summarize_sweep.py $O/${TAG}_sw10_stats 10000 33 $O/summ/${TAG}_sweep_10k.md || echo "(summary sweep 10k failed)"
  python3 tools/summarize_sweep.py $O/${TAG}_sw50_stats 50000 11 $O/summ/${TAG}_sweep_50k.md || echo "(summary sweep 50k failed)"
  python3 tools/summarize_sweep_pmc.py $O/summ/${TAG}_sweep_traffic.json $COMMIT 10000:$O/${TAG}_sw10_fetch:$O/${TAG}_sw10_write:7 50000:$O/${TAG}_sw50_fetch:$O/${TAG}_sw50_write:7 || echo "(summary sweep pmc failed)"
fi
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
echo profiles done
