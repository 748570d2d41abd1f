#!/usr/bin/env python3
"""Full-launch cross-attention only (64 sequences x 12 heads x 1500 keys, every row live), back to back over 12 different K / V^T images: the
one kernel and nothing else, so that a `rocprofv3 --pmc FETCH_SIZE` pass over this script gives the
HBM bytes of ONE FULL launch (bench.py's step mixes full and partly finished batches).  usage: python tools/xattn_pmc.py [B] [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 48
m = engine.Model(synth_model("small"))
ctx = engine.Context(m, max_batch=1)
ctx.set_precision("f16_mfma")
L = engine.lib()
L.skw_debug_xattn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
us, us_clk = C.c_float(), C.c_float()
assert L.skw_debug_xattn(ctx.h, B, 12, 0, iters, C.byref(us), C.byref(us_clk)) == 0
mb = B * 1500 * 768 * 2 * 2 / 1e6
print("B = %d: %.1f MB of K + V^T per launch (algorithmic), %.2f us per launch by HIP events = %.2f TB/s; %.2f us by the in-kernel clock (first wave in -> last wave out)" % (B, mb, us.value, mb / us.value, us_clk.value))
