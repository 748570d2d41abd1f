#!/bin/bash
# A/B of k_attn_encoder16's K-image swizzle (round 3's vs the conflict-free one) and of k_gemm16w against k_gemm16: same box, alternating, bench.py's own phase timings.
# usage (GPU box): bash tools/attn_ksw_ab.sh > gpurun_out/TAG_ab.txt
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2; do
  for v in "SKW_ATTN_KSW_R3=0" "SKW_ATTN_KSW_R3=1" "SKW_GEMM16W=0"; do
    env $v python3 bench.py --no-tts --steps 6 --warmup 2 --no-cpu-baseline --no-plugin-path --no-other-mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernels']
print('$v', 'value %.0f' % d['value'], 'encode %.2f' % d['modes']['f16_mfma']['encode_ms'], 'attn %.2f ms' % k['k_attn_encoder']['ms'], 'gemm %.2f ms' % k['k_gemm']['ms'])"
  done
done
