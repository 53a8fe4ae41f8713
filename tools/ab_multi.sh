#!/bin/bash
# round-robin of tools/step_time.py over the product and the named variants: tools/ab_multi.sh "v1 v2 ..." B [rounds]
set -e
cd $GRAFT_REPO_ROOT
VS=$1; B=${2:-1024}; R=${3:-2}
for r in $(seq $R); do
  echo "== product"; python3 tools/step_time.py $B
  for v in $VS; do echo "== $v"; VTC_HIP_LIB=$GRAFT_REPO_ROOT/vtc_amd/lib/variants/libvtc_$v.so python3 tools/step_time.py $B; done
done
