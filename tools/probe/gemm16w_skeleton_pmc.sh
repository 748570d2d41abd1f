export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$tag -o a -- python3 $R/tools/probe/gemm16w_skeleton_pmc.py > $O/skel_$tag.log 2>&1 || exit 1
  python3 - <<PY >> $O/skeleton_pmc.txt
import csv, glob, collections
for f in glob.glob("/tmp/pmc_$tag/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "k_gemm16w" in r["Kernel_Name"]]
    # dispatch order -> configuration: 3 warm-up + 4 timed launches per configuration
    by = collections.OrderedDict()
    for r in rows: by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    for ci in range(len(ids) // 7):
        grp = ids[ci * 7 + 3: ci * 7 + 7]
        names = sorted(by[grp[0]])
        print("config %d: " % ci + "  ".join("%s=%.4g" % (n, sum(by[i][n] for i in grp) / len(grp)) for n in names))
PY
done
cat $O/skeleton_pmc.txt
