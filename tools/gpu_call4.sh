#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py tests/test_gpu_plugin.py -x -q > $O/c4_tests.log 2>&1; rc=$?; tail -3 $O/c4_tests.log; [ $rc -eq 0 ] || exit 1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode"
timeout -k 10 200 $B > $O/c4_bench.json 2> $O/c4_bench.err && tail -c 300 $O/c4_bench.json
