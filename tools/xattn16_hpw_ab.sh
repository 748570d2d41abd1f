#!/bin/bash
# same-box A/B, two decode row groups (the default): cross attention with three heads per workgroup (128 workgroups per 32-row launch) against one head per workgroup (384)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for g in 3 2 1; do
  SKW_XATTN16_HPW=$g python3 bench.py --no-tts --steps 8 --warmup 3 --no-cpu-baseline --no-plugin-path --no-other-mode --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']
print('heads per workgroup $g round $r: %.2f ms per step  decode %.2f' % (d['ms_per_step'], m['decode_ms']))"
done; done
