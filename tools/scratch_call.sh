#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/c15_tests.log 2>&1; rc=$?; tail -3 $O/c15_tests.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/c15_smoke.log 2>&1; tail -1 $O/c15_smoke.log
timeout -k 10 400 python bench.py > $O/c15_bench.json 2> $O/c15_bench.err; tail -c 300 $O/c15_bench.json
