#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r04_lds $O/r04_lds2
rocprofv3 --list-avail 2>/dev/null | grep -i -E "lds|LDS" | head -40 > $O/lds_counters.txt
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/r04_lds -- python3 $R/tools/gemm_super_pmc.py > $O/r04_lds.log 2>&1 || echo "(pass 1 non-zero exit)"
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/r04_lds2 -- python3 $R/tools/gemm_super_pmc.py > $O/r04_lds2.log 2>&1 || echo "(pass 2 non-zero exit)"
cd $R
python3 tools/pmc_table.py gemm_phased $O/r04_lds $O/r04_lds2 | tail -24
