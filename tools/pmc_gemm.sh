#!/bin/bash
# PMC passes over the encoder kernels (k_gemm<*>, k_attn_encoder_v3): where do the SIMD cycles go?
export TMPDIR=/tmp; cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-tts --steps 1 --warmup 0 --clips 16 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag.log 2>&1
  python3 - <<PY
import csv, collections, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_encoder" in n or "k_gemm<" in n:
            k = (n.split("(")[0][-30:], r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(k, "n=%d avg=%.4g" % (v[0], v[1] / v[0]))
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag
done
