#!/bin/bash
# HBM bytes of one FULL cross-attention launch (every row live): rocprofv3 --pmc FETCH_SIZE over tools/xattn_pmc.py, program directly after `--`.
# usage (on the GPU box): bash tools/pmc_xattn.sh <tag>   -> gpurun_out/<tag>_xattn_full_launch_pmc.txt
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcx_$TAG -o p -- python3 $R/tools/xattn_pmc.py 64 48 > $O/${TAG}_xattn_pmc_run.log 2>&1 || exit 1
python3 - <<PY > $O/${TAG}_xattn_full_launch_pmc.txt
import csv
v = [float(r["Counter_Value"]) * 1024.0 * 2.0 for r in csv.DictReader(open("/tmp/pmcx_$TAG/p_counter_collection.csv")) if r.get("Counter_Name") == "FETCH_SIZE" and "k_dec_cross_attn" in r["Kernel_Name"]]
alg = 64 * 1500 * 768 * 2 * 2
print("k_dec_cross_attn, 64 rows, all live: %d launches under --pmc FETCH_SIZE (KiB -> bytes, x2 for gfx950's 128-B requests tallied at 64 B)" % len(v))
print("read per launch: mean %.2f MB, min %.2f, max %.2f; algorithmic %.2f MB (K + V^T, f16); ratio %.3f" % (sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6, alg / 1e6, sum(v) / len(v) / alg))
PY
cat $O/${TAG}_xattn_pmc_run.log | tail -1; cat $O/${TAG}_xattn_full_launch_pmc.txt
