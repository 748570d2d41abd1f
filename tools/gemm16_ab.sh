# A/B on one box: the K loop as one basic block (default) against the loop with the staging branch (SKW_GEMM16_OLDLOOP=1)
for i in 1 2; do
  for o in 0 1; do echo "== OLDLOOP=$o"; SKW_GEMM16_OLDLOOP=$o python tools/gemm16_probe.py 2>&1 | cut -c1-72 | tail -5; done
done
for o in 0 1 0 1; do SKW_GEMM16_OLDLOOP=$o python bench.py --no-tts --no-other-mode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLDLOOP=$o', j['value'], j['modes']['f16_mfma'])"; done
