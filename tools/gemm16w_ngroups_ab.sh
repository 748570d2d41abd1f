#!/bin/bash
# same-box A/B of k_gemm16w's feature-split tile walk (SKW_GEMM16W_NGROUPS=1 off | 0 automatic: FC1 in two n-tile groups), alternating, three rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for g in 1 0; do
  SKW_GEMM16W_NGROUPS=$g python3 bench.py --no-tts --steps 8 --warmup 3 --no-cpu-baseline --no-plugin-path --no-other-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = d['modes']['f16_mfma']; k = d['roofline']['kernels']
print('ngroups $g round $r: %.2f ms per step  encode %.2f  decode %.2f  k_gemm %.2f ms %s TF/s' % (d['ms_per_step'], m['encode_ms'], m['decode_ms'], k['k_gemm']['ms'], k['k_gemm'].get('tflops')))"
done; done
