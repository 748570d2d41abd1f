set -e
for cfg in "3 3 1 0" "3 3 1 1" "4 3 1 1" "2 3 1 1" "3 3 0 1" "3 1 1 1"; do
  set -- $cfg
  echo "== RD=$1 HPW=$2 FRAG=$3 IL=$4"
  SKW_XATTN16=1 SKW_XATTN16_RD=$1 SKW_XATTN16_HPW=$2 SKW_XATTN16_FRAG=$3 SKW_XATTN16_IL=$4 python tools/xattn_probe.py 64 2>&1 | grep -E "probe  0" | tail -1
done
