#!/bin/bash
# round-2 session-2 GPU call: parity of the fused-query cross attention + A/B timings
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py -x -q > $O/c1_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c1_tests.log
tail -3 $O/c1_tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode"
timeout -k 10 200 $B > $O/c1_bench_fused.json 2> $O/c1_bench_fused.err && tail -c 600 $O/c1_bench_fused.json
SKW_XATTN_FUSEQ=0 timeout -k 10 200 $B > $O/c1_bench_unfused.json 2> $O/c1_bench_unfused.err && tail -c 600 $O/c1_bench_unfused.json
SKW_XATTN_NT=1 timeout -k 10 200 $B > $O/c1_bench_fused_nt.json 2> $O/c1_bench_fused_nt.err && tail -c 600 $O/c1_bench_fused_nt.json
SKW_XATTN_FUSEQ=0 SKW_XATTN_NT=1 timeout -k 10 200 $B > $O/c1_bench_unfused_nt.json 2> $O/c1_bench_unfused_nt.err && tail -c 600 $O/c1_bench_unfused_nt.json
