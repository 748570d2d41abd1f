#!/bin/bash
# same-box A/B of the decode row-group count (SKW_DECODE_GROUPS=1|2), alternating, three rounds: whole-step ms and decode ms per configuration
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for g in 1 2; do
  SKW_DECODE_GROUPS=$g python3 bench.py --no-tts --steps 8 --warmup 3 --no-cpu-baseline --no-plugin-path --no-other-mode --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']
print('groups $g round $r: %.2f ms per step  encode %.2f  decode %.2f  -> %.0fx' % (d['ms_per_step'], m['encode_ms'], m['decode_ms'], d['value']))"
done; done
