#!/bin/bash
# same-box A/B of the encoder attention's register cap (SKW_ATTN_OCC=2: 143 registers, three workgroups per CU | 4: 128 registers + 13 dwords of scratch, four per CU)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for g in 2 4; do
  SKW_ATTN_OCC=$g python3 bench.py --no-tts --steps 6 --warmup 2 --no-cpu-baseline --no-plugin-path --no-other-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']; k = d['roofline']['kernels']
print('attention occupancy bound $g round $r: %.2f ms per step  encode %.2f  k_attn_encoder %.3f ms (%s TF/s)' % (d['ms_per_step'], m['encode_ms'], k['k_attn_encoder']['ms'], k['k_attn_encoder'].get('tflops')))"
done; done
