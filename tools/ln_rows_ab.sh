#!/bin/bash
# same-box A/B of the encoder LayerNorm's rows per wave (SKW_LN_ROWS=1 | 2), alternating, three rounds: k_layernorm ms per batch from bench.py's per-kernel event pairs
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for g in 1 2; do
  SKW_LN_ROWS=$g python3 bench.py --no-tts --steps 6 --warmup 2 --no-cpu-baseline --no-plugin-path --no-other-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']; k = d['roofline']['kernels']
print('ln rows $g round $r: %.2f ms per step  encode %.2f  k_layernorm %.3f ms in %d launches' % (d['ms_per_step'], m['encode_ms'], k['k_layernorm']['ms'], k['k_layernorm']['launches']))"
done; done
