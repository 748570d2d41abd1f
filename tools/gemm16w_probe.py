#!/usr/bin/env python3
"""The encoder's GEMM shapes through k_gemm16 (256 x 256 tiles, one workgroup per CU, both operands through LDS) and through k_gemm16w (128 x 256 tiles, two
workgroups per CU, weights from their fragment-order image straight into MFMA operands), same process, interleaved rounds.  usage: python tools/gemm16w_probe.py [rounds]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine

m = engine.Model(synth_model("small"))
ctx = engine.Context(m, max_batch=1)
L = engine.lib()
L.skw_debug_gemm16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
EPI = {"F32": 0, "GELU_KPERM": 2, "HEADS": 4, "VT": 5, "PLAIN": 6}
shapes = [("Q/K proj", 96000, 768, 768, "HEADS"), ("V^T", 96000, 768, 768, "VT"), ("cross K", 96000, 768, 768, "PLAIN"), ("O proj", 96000, 768, 768, "F32"),
          ("FC1", 96000, 3072, 768, "GELU_KPERM"), ("FC2", 96000, 768, 3072, "F32")]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2 and sys.argv[2] == "parts":      # k_gemm16w with parts switched off: where a launch's time goes
    NOEPI, ONEWG, NOMFMA, NOW, NOA = 1024, 2048, 4096, 8192, 16384
    cols = [("full", 0), ("1 WG/CU", ONEWG), ("no epi", NOEPI), ("no epi 1WG", NOEPI | ONEWG), ("no MFMA", NOMFMA), ("no MFMA no epi", NOMFMA | NOEPI), ("no W loads", NOW | NOEPI), ("no A DMA", NOA | NOEPI), ("epi only", NOMFMA | NOW | NOA), ("no glob epi", 65536), ("skeleton", NOMFMA | NOW | NOA | NOEPI), ("A in L2", 32768), ("A in L2 no epi", 32768 | NOEPI), ("A in L2 no epi no MFMA", 32768 | NOEPI | NOMFMA)]
    print("%-10s " % "" + " ".join("%14s" % c[0][:14] for c in cols))
    for name, M, N, K, epi in shapes:
        row = []
        for cname, bits in cols:
            best = 1e9
            for r in range(rounds):
                ms = C.c_float()
                assert L.skw_debug_gemm16(ctx.h, M, N, K, EPI[epi], 512 | bits, 10, C.byref(ms)) == 0
                best = min(best, ms.value * 1e3)
            row.append(best)
        print("%-10s " % name + " ".join("%14.1f" % x for x in row))
    sys.exit(0)
print("%-10s %-30s %12s %12s %8s" % ("", "M x N x K / epilogue", "k_gemm16 us", "k_gemm16w us", "TF/s (w)"))
tot = [0.0, 0.0]
for name, M, N, K, epi in shapes:
    t = {0: [], 512: []}
    for r in range(rounds):
        for probe in (0, 512):
            ms = C.c_float()
            assert L.skw_debug_gemm16(ctx.h, M, N, K, EPI[epi], probe, 10, C.byref(ms)) == 0, engine.lib().skw_ctx_last_error(ctx.h)
            t[probe].append(ms.value * 1e3)
    a, b = min(t[0]), min(t[512])
    w = {"Q/K proj": 2, "FC1": 1, "FC2": 1, "O proj": 1, "V^T": 2, "cross K": 1}[name]
    tot[0] += a * w; tot[1] += b * w
    print("%-10s %-30s %12.1f %12.1f %8.0f   (all rounds: %s | %s)" % (name, "%d x %d x %d / %s" % (M, N, K, epi), a, b, 2.0 * M * N * K / b / 1e6,
          " ".join("%.0f" % x for x in t[0]), " ".join("%.0f" % x for x in t[512])))
print("per encoder layer (Q, K, V, O, FC1, FC2) + one cross K / V^T pair: k_gemm16 %.0f us, k_gemm16w %.0f us" % (tot[0], tot[1]))
