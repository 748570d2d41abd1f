set -e
for cfg in "0 0" "147 0" "74 0" "0 147" "147 80" "147 147"; do
  set -- $cfg
  echo "== touch K $1 MB, V $2 MB"
  SKW_XATTN_PROBE_TOUCH_K=$1 SKW_XATTN_PROBE_TOUCH_V=$2 python tools/xattn_probe.py 64 2>&1 | grep -E "f16_mfma|probe  0|probe  1" | tail -3
done
