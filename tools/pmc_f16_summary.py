#!/usr/bin/env python3
"""Derived figures from tools/pmc_f16.sh's counter averages (gpurun_out/TAG_pmc_sq_f16.txt): matrix-core busy share, LDS bank-conflict share, VALU share.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32_f16: it equals 16 x SQ_INSTS_MFMA here);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  usage: python tools/pmc_f16_summary.py FILE"""
import collections
import sys

vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) < 4 or not p[-1].startswith("avg="):
        continue
    name = " ".join(p[:-3]); ctr = p[-3]
    vals[name][ctr] = float(p[-1][4:])
print("%-30s %10s %12s %12s %14s %12s" % ("kernel", "us/launch", "clock GHz*", "MFMA busy", "LDS conflict", "VALU active"))
for k, v in sorted(vals.items()):
    if "GRBM_GUI_ACTIVE" not in v or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                       # kernel duration in shader-engine clocks
    mfma = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc     # share of every SIMD's cycles with the matrix core busy
    lds = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_ACTIVE_INST_LDS", 0.0), 1.0)
    valu = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / 1024.0 / cyc
    print("%-30s %10s %12s %11.1f%% %13.1f%% %11.1f%%" % (k, "-", "-", 100 * mfma, 100 * lds, 100 * valu))
print("* durations and clocks: see the kernel trace summary of the same round; SQ_ACTIVE_INST_* count quad-cycles (x4), per the guide's PMC unit table")
