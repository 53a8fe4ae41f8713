#!/bin/bash
# alternating runs of tools/step_time.py on two builds: the product and vtc_amd/lib/variants/libvtc_$1.so
set -e
cd $GRAFT_REPO_ROOT
V=${1:-base}; shift || true
B=${@:-1024}
for r in 1 2 3; do
  echo "== product"; python3 tools/step_time.py $B
  echo "== $V"; VTC_HIP_LIB=$GRAFT_REPO_ROOT/vtc_amd/lib/variants/libvtc_$V.so python3 tools/step_time.py $B
done
