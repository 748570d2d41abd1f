#!/usr/bin/env python3
"""Where does a decode-step cross-attention launch (k_dec_cross_attn, 64 sequences x 12 heads x 1500 keys, 295 MB of K / V^T) spend its time?
Back-to-back launches over 12 different K / V^T images with parts switched off: 1 = no P.V MFMAs, 2 = no score chains, 4 = no LDS transposes
either, 8 = no V^T loads (skw_debug_xattn).  usage: python tools/xattn_probe.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = engine.Model(synth_model("small"))
ctx = engine.Context(m, max_batch=1)
L = engine.lib()
L.skw_debug_xattn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
mb = B * 1500 * 768 * 2 * 2 / 1e6
for prec in ("exact", "f16_mfma"):
  ctx.set_precision(prec)
  print("B = %d, %s: %.1f MB of K + V^T per launch" % (B, prec, mb))
  for probe, what in ((0, "whole kernel"), (1, "no P.V MFMAs"), (2, "no score chains"), (3, "neither"), (6, "no LDS transposes, no chains"), (7, "loads + softmax only"), (8, "no V^T phase"), (14, "K loads only")):
    us = C.c_float()
    if prec == "f16_mfma" and probe not in (0, 1): continue
    assert L.skw_debug_xattn(ctx.h, B, 12, probe, 240, C.byref(us)) == 0, ctx.last_error() if hasattr(ctx, "last_error") else "error"
    print("probe %2d  %-34s %7.2f us  %6.2f TB/s" % (probe, what, us.value, mb / us.value * (0.5 if probe & 8 else 1.0)))
