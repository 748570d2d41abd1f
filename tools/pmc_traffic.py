#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json (HBM bytes per launch per kernel class, read by bench.py for roofline.traffic) from two
rocprofv3 counter_collection CSVs: one --pmc FETCH_SIZE pass and one --pmc WRITE_SIZE pass over `bench.py --steps 1 --warmup 0`.
gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE tallies 128-B requests at 64 B -> x2.
usage: pmc_traffic.py fetch.csv write.csv out.json [precision]   (the result is stored under that precision's key; other keys of an existing out.json are kept)"""
import csv, collections, json, sys
CLASSES = [("k_mel", "k_mel"), ("k_gemm_smallm", "k_gemm_smallm"), ("k_gemm16_small", "k_gemm_smallm"), ("k_gemm16_vocab", "k_gemm_smallm"), ("k_gemm<", "k_gemm"), ("k_gemm16<", "k_gemm"), ("k_gemm16w<", "k_gemm"), ("k_attn_encoder", "k_attn_encoder"), ("k_layernorm", "k_layernorm"),
           ("k_dec_cross_attn", "k_dec_cross_attn"), ("k_dec_attn", "k_dec_self_attn"), ("k_dec_sample", "k_dec_sample")]
def cls(name):
    for pat, c in CLASSES:
        if pat in name: return c
    return None
def load(path, counter, scale):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter: continue
        c = cls(r["Kernel_Name"])
        if c: agg[c][0] += 1; agg[c][1] += float(r["Counter_Value"]) * 1024.0 * scale
    return agg
f = load(sys.argv[1], "FETCH_SIZE", 2.0); w = load(sys.argv[2], "WRITE_SIZE", 1.0)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0`; KiB -> bytes; FETCH_SIZE doubled "
               "(gfx950: 128-B requests tallied at 64 B); Infinity-Cache hits are included by these counters", "per_launch_bytes": {}}
for c in f:
    n = f[c][0]; rd = f[c][1] / n; wr = w[c][1] / max(1, w[c][0]) if c in w else 0.0
    out["per_launch_bytes"][c] = {"launches": n, "read": int(rd), "write": int(wr), "total": int(rd + wr)}
import os
prec = sys.argv[4] if len(sys.argv) > 4 else "f16_mfma"
allp = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
if "per_launch_bytes" in allp: allp = {"exact": allp}          # round-1 layout: one precision
allp[prec] = out
json.dump(allp, open(sys.argv[3], "w"), indent=1)
for c, v in sorted(out["per_launch_bytes"].items(), key=lambda kv: -kv[1]["total"] * kv[1]["launches"]):
    print("%-18s launches %6d  read %10.2f MB  write %9.2f MB per launch" % (c, v["launches"], v["read"] / 1e6, v["write"] / 1e6))
