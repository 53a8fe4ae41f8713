#!/bin/bash
# round-4 rocprofv3 evidence, run on the GPU box from the repo root (gpurun): stats + PMC passes of the bench command (towers
# serialised), the super-row PMC table of VERDICT r3 #3, a steady-state kernel trace.  Outputs under gpurun_out/r04_*.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
# (the one-launch CAM is an ordinary launch by default; VTC_CAM_COOP=1 -- cooperative -- makes rocprofv3 of ROCm 7.2 segfault
# in its teardown, after writing its output: tools/exit_probe.py)
export VTC_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-extra --no-cpu --no-sweep --no-independence > $O/r04_stats.json 2> $O/r04_stats.err || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r04_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r04_fetch.err || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r04_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu --no-sweep --no-independence > /dev/null 2> $O/r04_write.err || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
unset VTC_OVERLAP
for s in 1 2 4 8 16; do
  export VTC_GEMM_SUPER=$s
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r04_super_$s -- python3 $R/tools/gemm_super_pmc.py > $O/r04_super_$s.log 2>&1 || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
done
unset VTC_GEMM_SUPER
for c in 1 3; do
  export VTC_GEMM_CG=$c
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r04_cg_$c -- python3 $R/tools/gemm_super_pmc.py > $O/r04_cg_$c.log 2>&1 || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
done
unset VTC_GEMM_CG
rocprofv3 --kernel-trace --output-format csv -d $O/r04_step -- python3 $R/tools/step_kernels.py 64 4 > $O/r04_step.log 2>&1 || echo "(non-zero exit: see the .err file; rocprofv3 has written its output before the process teardown)"
cd $R
find $O/r04_stats -name "*kernel_trace.csv" -size +20M -delete
for s in 1 2 4 8 16; do grep "us per launch" $O/r04_super_$s.log; python tools/pmc_table.py gemm_phased $O/r04_super_$s | tail -2; done
for c in 1 3; do grep "us per launch" $O/r04_cg_$c.log; python tools/pmc_table.py gemm_phased $O/r04_cg_$c | tail -2; done
echo profiles done
