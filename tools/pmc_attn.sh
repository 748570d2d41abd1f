export TMPDIR=/tmp; cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$tag -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-tts --steps 1 --warmup 0 --clips 16 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$tag.log 2>&1
  python3 - <<PY
import csv, collections, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$tag/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "attn_encoder" in r["Kernel_Name"] or "k_gemm<2>" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(k, "n=%d avg=%.4g" % (v[0], v[1] / v[0]))
PY
done
