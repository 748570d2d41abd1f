#!/bin/bash
# Kernel trace of a short f16_mfma bench run under the given environment, reduced to a per-(kernel, grid) summary and one decode step's timeline.
# usage (on the GPU box): [ENV=... ] bash tools/decode_timeline.sh <tag>
TAG=${1:-tl}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o r -- python3 $R/bench.py --no-tts --steps 2 --warmup 1 --precision f16_mfma --no-other-mode --no-cpu-baseline --no-roofline > $O/${TAG}_bench.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py /tmp/tl_$TAG/r_kernel_trace.csv 30 > $O/${TAG}_summary.txt
python3 $R/tools/step_timeline.py /tmp/tl_$TAG/r_kernel_trace.csv 150 > $O/${TAG}_timeline.txt 2>&1
tail -1 $O/${TAG}_bench.log | cut -c1-400; tail -1 $O/${TAG}_timeline.txt
