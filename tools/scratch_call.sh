#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_f16.py -x -q > $O/c11_tests.log 2>&1; rc=$?; tail -3 $O/c11_tests.log; [ $rc -eq 0 ] || exit 1
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-mode"
timeout -k 10 200 $B > $O/c11_bench.json 2> $O/c11_bench.err && tail -c 200 $O/c11_bench.json
