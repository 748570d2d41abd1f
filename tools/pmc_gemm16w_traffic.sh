#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per k_gemm16w<EPI> variant and grid (separate --pmc passes, --kernel-trace only, program directly after `--`): which of the encoder's GEMM
# shapes over-fetch.  FETCH_SIZE in KiB, doubled per the guide's gfx950 correction (128-byte requests tallied at 64 B).  usage: bash tools/pmc_gemm16w_traffic.sh TAG
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
: > $O/${TAG}_gemm16w_traffic.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcg_$C -o p -- python3 $R/bench.py --no-tts --steps 1 --warmup 0 --no-other-mode --no-cpu-baseline --no-roofline --no-plugin-path > $O/${TAG}_pmcg_$C.log 2>&1 || { echo "pass $C failed" >> $O/${TAG}_gemm16w_traffic.txt; continue; }
  python3 - $C >> $O/${TAG}_gemm16w_traffic.txt <<'PY'
import csv, collections, glob, sys
C = sys.argv[1]
for f in glob.glob("/tmp/pmcg_%s/*counter_collection.csv" % C):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_gemm16w<" in n:
            k = (n[n.index("k_gemm16w<"):].split("(")[0][:24], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        mb = v[1] / v[0] * 1024.0 * (2.0 if C == "FETCH_SIZE" else 1.0) / 1e6
        print("%-11s %-26s grid %-8s launches %-4d %8.1f MB per launch" % (C, k[0], k[1], v[0], mb))
PY
done
cat $O/${TAG}_gemm16w_traffic.txt
