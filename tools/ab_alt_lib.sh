# A/B on one box: the in-tree libskw_engine.so against a second build of it (streamkit_amd/alt_libskw_engine.so, SKW_ENGINE_SO)
for o in 0 1 0 1; do
  if [ $o = 1 ]; then export SKW_ENGINE_SO=$PWD/streamkit_amd/alt_libskw_engine.so; else unset SKW_ENGINE_SO; fi
  python bench.py --no-tts --no-other-mode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ALT=$o', j['value'], j['modes']['f16_mfma'], {k:v.get('ms') for k,v in j['roofline']['kernels'].items()})"
done
