# A/B on one box: decode GEMM weight loads with the default / the non-temporal policy (second build: streamkit_amd/alt_libskw_engine.so, -DSKW_DEC_W_AUX=2)
for o in 0 1 0 1; do
  if [ $o = 1 ]; then export SKW_ENGINE_SO=$PWD/streamkit_amd/alt_libskw_engine.so; else unset SKW_ENGINE_SO; fi
  python bench.py --no-other-mode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W_NT=$o', j['value'], j['modes']['f16_mfma'], j['roofline']['avg_launch_ms'], {k:v.get('ms') for k,v in j['roofline']['kernels'].items()})"
done
