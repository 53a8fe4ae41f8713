#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/r04_b1
cd /tmp
export VTC_OVERLAP=${VTC_OVERLAP:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_b1 -- python3 $R/tools/step_time.py 1 > $O/r04_b1.log 2>&1 || echo "(non-zero exit)"
cd $R
f=$(find $O/r04_b1 -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
nf=max(int(r['Calls']) for r in rows if 'cam_fused' in r['Name'])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"forwards {nf}; kernel time per forward {tot/nf/1e3:.1f} us")
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls/fwd {int(r['Calls'])/nf:6.1f} avg {float(r['AverageNs'])/1e3:7.2f} us  per fwd {float(r['TotalDurationNs'])/nf/1e3:7.1f} us")
PY
find $O/r04_b1 -name "*kernel_trace.csv" -delete
