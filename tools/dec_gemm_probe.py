#!/usr/bin/env python3
"""Where does a decode-step GEMM launch (k_gemm16_small, 64 rows) spend its time?  Chains of dependent launches of the decoder's shapes
with parts switched off (skw_debug_gemm16 with M <= 64: 1 = no weight loads, 2 = no activation loads, 4 = no exchange/epilogue, 8 = no stores, 32 = weights as a fragment-order image, 64 = activations addressed as one; for the vocabulary kernel 1 / 2 / 8 likewise and 16 = the strip kernel instead).
usage: python tools/dec_gemm_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine

m = engine.Model(synth_model("small"))
ctx = engine.Context(m, max_batch=1)
ctx.set_precision("f16_mfma")
L = engine.lib()
L.skw_debug_gemm16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
EPI = {"F32": 0, "GELU_KPERM": 2, "PLAIN": 6, "QKV": 8}
shapes = [("out proj", 64, 768, 768, "F32"), ("q proj", 64, 768, 768, "PLAIN"), ("qkv", 64, 2304, 768, "QKV"), ("fc1", 64, 3072, 768, "GELU_KPERM"), ("fc2", 64, 768, 3072, "F32"), ("logits", 64, 51865, 768, "F32"),
          ("out proj 32", 32, 768, 768, "F32"), ("fc1 16", 16, 3072, 768, "GELU_KPERM")]
probes = (0, 1, 2, 3, 4, 7, 8, 9, 11, 16, 32, 34, 96)
print("%-12s %-22s" % ("", "M x N x K / epilogue") + "".join(" %8s" % ("p=%d" % p) for p in probes) + "   (us per dependent launch)")
for name, M, N, K, epi in shapes:
    t = []
    for probe in probes:
        ms = C.c_float()
        assert L.skw_debug_gemm16(ctx.h, M, N, K, EPI[epi], probe, 200, C.byref(ms)) == 0
        t.append(ms.value * 1e3)
    print("%-12s %-22s" % (name, "%d x %d x %d / %s" % (M, N, K, epi)) + "".join(" %8.2f" % x for x in t))
