#!/usr/bin/env python3
"""Where does a k_gemm16 launch spend its time?  Times the encoder's GEMM shapes with parts of the kernel switched off
(skw_debug_gemm16: bit 0 no K-loop DMA, bit 1 no MFMAs, bit 2 no epilogue, bit 3 no global stores, bit 4 no per-element epilogue math).  usage: python tools/gemm16_probe.py [small]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine

m = engine.Model(synth_model(sys.argv[1] if len(sys.argv) > 1 else "small"))
ctx = engine.Context(m, max_batch=1)
L = engine.lib()
L.skw_debug_gemm16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
EPI = {"F32": 0, "GELU_KPERM": 2, "HEADS": 4, "PLAIN": 6}
shapes = [("Q/K proj", 96000, 768, 768, "HEADS"), ("cross K", 96000, 768, 768, "PLAIN"), ("O proj", 96000, 768, 768, "F32"), ("FC1", 96000, 3072, 768, "GELU_KPERM"), ("FC2", 96000, 768, 3072, "F32")]
print("%-10s %-28s %9s %9s %9s %9s %9s %9s %9s" % ("", "M x N x K / epilogue", "full us", "TF/s", "no DMA", "no MFMA", "no epi", "no store", "no math"))
for name, M, N, K, epi in shapes:
    t = {}
    for probe in (0, 1, 2, 4, 8, 16, 32, 64, 4 | 32, 4 | 64, 256, 256 | 4):
        ms = C.c_float()
        assert L.skw_debug_gemm16(ctx.h, M, N, K, EPI[epi], probe, 10, C.byref(ms)) == 0
        t[probe] = ms.value * 1e3
    print("%-10s %-28s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f   nt A %6.1f  nt W %6.1f  loop only: nt A %6.1f  nt W %6.1f" % (name, "%d x %d x %d / %s" % (M, N, K, epi), t[0], 2.0 * M * N * K / t[0] / 1e6, t[1], t[2], t[4], t[8], t[16], t[32], t[64], t[36], t[68]) + "   A from L2: %6.1f  loop only %6.1f" % (t[256], t[260]))
