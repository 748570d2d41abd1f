#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_f16.py -x -q > $O/c5_tests.log 2>&1; rc=$?; tail -3 $O/c5_tests.log; [ $rc -eq 0 ] || exit 1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode"
timeout -k 10 200 $B > $O/c5_bench.json 2> $O/c5_bench.err && tail -c 200 $O/c5_bench.json
SKW_DEC_RD_DEEP=12 timeout -k 10 200 $B > $O/c5_bench_rd12.json 2> $O/c5_bench_rd12.err && tail -c 200 $O/c5_bench_rd12.json
