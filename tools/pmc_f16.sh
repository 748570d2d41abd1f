#!/bin/bash
# SQ counter passes over the f16_mfma precision's dominant kernels (k_gemm16<*>, k_attn_encoder16, k_dec_cross_attn): matrix-core busy cycles
# against the kernel's busy cycles, VALU / LDS activity and bank conflicts.  Separate --pmc passes with --kernel-trace only; program directly after `--`.
# usage (on the GPU box): bash tools/pmc_f16.sh TAG   -> gpurun_out/TAG_pmc_sq_f16.txt
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
: > $O/${TAG}_pmc_sq_f16.txt
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-48)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc16_$tag -o a -- python3 $R/bench.py --no-tts --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-mode > $O/${TAG}_pmc16_$tag.log 2>&1 || { echo "pass $C failed" >> $O/${TAG}_pmc_sq_f16.txt; continue; }
  python3 - "$tag" >> $O/${TAG}_pmc_sq_f16.txt <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
for f in glob.glob("/tmp/pmc16_%s/*counter_collection.csv" % tag):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("k_gemm16w<", "k_gemm16<", "k_attn_encoder16", "k_dec_cross_attn", "k_gemm16_small<0", "k_layernorm"):
            if key in n:
                short = n.split("(")[0]
                if key in ("k_gemm16<", "k_gemm16w<"): short = short[short.index(key):][:28]
                else: short = key
                k = (short, r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"]); break
    for k, v in sorted(agg.items()): print("%-30s %-28s launches=%-6d avg=%.6g" % (k[0], k[1], v[0], v[1] / v[0]))
PY
  rm -rf /tmp/pmc16_$tag
done
cat $O/${TAG}_pmc_sq_f16.txt
