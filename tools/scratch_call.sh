#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "log_mel or contract or golden or encoder_stages" > $O/c22_tests.log 2>&1; rc=$?; tail -2 $O/c22_tests.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-mode > $O/c22_bench.json 2> $O/c22_bench.err; tail -c 100 $O/c22_bench.json
