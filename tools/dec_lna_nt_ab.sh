#!/bin/bash
# same-box A/B, two decode row groups (32-row launches): strips per workgroup of the LayerNorm-folding decode GEMMs with N >= 2048 (QKV, FC1): SKW_DEC_LNA_NT = 4 (default) | 2 | 1
cd $GRAFT_REPO_ROOT
for r in 1 2; do for g in 4 2 1; do
  SKW_DEC_LNA_NT=$g python3 bench.py --no-tts --steps 8 --warmup 3 --no-cpu-baseline --no-plugin-path --no-other-mode --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']
print('lnA strips per workgroup $g round $r: %.2f ms per step  decode %.2f' % (d['ms_per_step'], m['decode_ms']))"
done; done
