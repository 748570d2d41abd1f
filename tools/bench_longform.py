#!/usr/bin/env python3
"""Long-form Oneshot transcription (multi-window clips: every window after the first carries up to 224 tokens of the clip's earlier text in its prompt):
Whisper-small, B clips of S seconds, f16_mfma, with the prompt as one multi-row pass (default) and with one prompt token per step (SKW round-2 behaviour).
usage: python tools/bench_longform.py [B] [seconds]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from streamkit_amd import engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = float(sys.argv[2]) if len(sys.argv) > 2 else 95.0
m = engine.Model(synth_model("small"))
ctx = engine.Context(m, max_batch=B, max_samples=int(16000 * S) + 16)
ctx.set_precision("f16_mfma")
L = engine.lib(); L.skw_debug_set_prompt_pass.argtypes = [C.c_void_p, C.c_int]
pcms = [synth.clip(200 + c, int(16000 * S)) for c in range(B)]
p = ctx.default_params(); p.suppress_nst = 1
ref = None
for on in (1, 0, 1, 0):
    L.skw_debug_set_prompt_pass(ctx.h, on)
    t0 = time.perf_counter(); res = ctx.full_batch(pcms, p); dt = time.perf_counter() - t0
    t = ctx.timing()
    ids = [[x[0] for x in r["tokens"]] for r in res]
    ref = ref or ids
    print("%d x %.0f s, prompt %s: %.1f ms wall = %.0fx real time; encode %.1f ms, decode %.1f ms, %d windows, %d decoder passes, %d tokens%s"
          % (B, S, "in one pass      " if on else "one token per step", dt * 1e3, B * S / dt, t["encode_ms"], t["decode_ms"], t["n_windows"], t["n_decode_steps"], t["n_tokens"],
             "" if ids == ref else "  TOKENS DIFFER"))
