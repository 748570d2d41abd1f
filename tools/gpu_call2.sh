#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py -x -q > $O/c2_tests.log 2>&1; rc=$?; tail -3 $O/c2_tests.log; [ $rc -eq 0 ] || exit 1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode"
timeout -k 10 200 $B > $O/c2_bench.json 2> $O/c2_bench.err && tail -c 300 $O/c2_bench.json
SKW_GEMM16_STAGGER=1 timeout -k 10 200 $B > $O/c2_bench_stagger.json 2> $O/c2_bench_stagger.err && tail -c 300 $O/c2_bench_stagger.json
bash tools/pmc_f16.sh r02f
