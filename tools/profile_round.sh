#!/bin/bash
# Reproduce the evidence under profiles/ for one round tag: kernel-trace stats OF THE TIMED CONFIGURATION (the driver's --steps 20 --warmup 3, nothing but the timed steps: no roofline
# instrument, no plugin legs — so that a kernel's average is the average of the launches bench.py's line describes), per-(kernel, grid) summary, decode-step timeline, bench line,
# PMC HBM traffic (separate --pmc passes, program directly after `--`).
# usage (on the GPU box): bash tools/profile_round.sh r02a [f16_mfma|exact]   -> writes gpurun_out/<tag>_*; copy what you want judged into profiles/
TAG=${1:-rXX}; PREC=${2:-f16_mfma}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r -- python3 $R/bench.py --no-tts --steps 20 --warmup 3 --precision $PREC --no-other-mode --no-cpu-baseline --no-roofline --no-plugin-path > $O/${TAG}_bench_under_rocprof.log 2>&1 || exit 1
cp /tmp/prof_$TAG/r_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
python3 $R/tools/prof_summary.py /tmp/prof_$TAG/r_kernel_trace.csv 40 > $O/${TAG}_kernel_trace_summary.txt
python3 $R/tools/step_timeline.py /tmp/prof_$TAG/r_kernel_trace.csv 150 > $O/${TAG}_decode_step_timeline.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o p -- python3 $R/bench.py --no-tts --steps 1 --warmup 0 --precision $PREC --no-other-mode --no-cpu-baseline --no-roofline --no-plugin-path > $O/${TAG}_pmc_$C.log 2>&1 || exit 1
done
cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json 2>/dev/null
python3 $R/tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_${TAG}_WRITE_SIZE/p_counter_collection.csv $O/${TAG}_pmc_traffic.json $PREC > $O/${TAG}_pmc_hbm_traffic_per_kernel.txt
cd $R && cp $O/${TAG}_pmc_traffic.json profiles/pmc_traffic.json && timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --precision $PREC > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
tail -1 $O/${TAG}_bench_line.json
