set -e
cd $GRAFT_REPO_ROOT
for cfg in "1 0" "0 0" "0 1" "0 3" "1 0"; do
  set -- $cfg
  echo "== VTC_LN_FOLD=$1 VTC_FUSED_ATTN=$2"
  VTC_LN_FOLD=$1 VTC_FUSED_ATTN=$2 python3 tools/step_time.py 1 8 50
done
