# the f16_mfma tests under each measured alternative (README: environment switches): every path the switches select stays green
set -o pipefail
for e in "SKW_DEC_LN_STATS=0" "SKW_XATTN_FRAG=0" "SKW_DEC_WFRAG=0" "SKW_DEC_AFRAG=0" "SKW_DEC_ATTN_FASTV=0" "SKW_PROMPT_PASS=0" "SKW_DECODE_GROUPS=2"; do
  echo "== $e"
  env $e python -m pytest tests/test_gpu_f16.py -m gpu -q -x -p no:cacheprovider -k "logits_within or ragged_batch or teacher_forced_ragged or cross_kv or other_widths or multi_window" 2>&1 | tail -2
done
