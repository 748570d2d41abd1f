#!/bin/bash
O=gpurun_out; mkdir -p $O
python tools/xattn_probe.py 64 > $O/c6_xattn_probe.txt 2>&1; cat $O/c6_xattn_probe.txt
timeout -k 10 400 python -m pytest tests/test_gpu_f16.py -x -q > $O/c6_tests.log 2>&1; rc=$?; tail -3 $O/c6_tests.log; [ $rc -eq 0 ] || exit 1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 $B > $O/c6_bench.json 2> $O/c6_bench.err && tail -c 200 $O/c6_bench.json
