#!/bin/bash
# alternating runs of tools/step_time.py under different environment settings: tools/ab_env.sh B "VAR=a" "VAR=b" ...
set -e
cd $GRAFT_REPO_ROOT
B=$1; shift
for r in 1 2; do
  for e in "$@"; do echo "== $e"; env $e python3 tools/step_time.py $B; done
done
