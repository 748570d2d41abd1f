# the one-pass cross attention alone (tools/xattn_probe.py, 64-row launches over 12 K / V^T images): policy, blocks in flight, block assignment, workgroup shape
for cfg in "1 3 1 3" "0 3 1 3" "1 2 1 3" "1 4 1 3" "1 3 0 3" "1 3 1 1" "1 3 1 3"; do
  set -- $cfg
  echo "== NT=$1 RD=$2 IL=$3 HPW=$4"
  SKW_XATTN16_NT=$1 SKW_XATTN16_RD=$2 SKW_XATTN16_IL=$3 SKW_XATTN16_HPW=$4 python tools/xattn_probe.py 64 2>&1 | grep -E "probe  0" | tail -1
done
