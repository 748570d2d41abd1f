#!/usr/bin/env python3
"""The Oneshot batch of bench.py (64 x 30 s, Whisper-small geometry) on a block-quantised file: ggml's q8 arithmetic (exact precision,
skw_kernels_q8.hip) beside the file's f16 twin in both precisions.  Informational: the headline benchmark is the f16 file (bench.py).
usage: python tools/bench_quant.py [q5_1] [clips=64] [--profile]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from streamkit_amd import engine, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "q5_1"
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 64
tool = os.path.join(ROOT, "tools", "make_synth_model")
if not os.path.exists(tool):
    subprocess.check_call(["gcc", "-O2", "-o", tool, tool + ".c", "-lm"])
src, qpath = "/tmp/skw_bq_small.bin", "/tmp/skw_bq_small_%s.bin" % kind
subprocess.check_call([tool, src, "--size", "small", "--seed", "1234"])
t0 = time.time()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "quantize_ggml.py"), src, qpath, kind])
print("quantised in %.1f s" % (time.time() - t0), flush=True)
pcms = [synth.clip(c, 480000) for c in range(B)]
for label, qm, prec in (("ggml q8 arithmetic, exact precision", 1, "exact"), ("f16 twin, exact precision", 0, "exact"), ("f16 twin, f16_mfma precision", 1, "f16_mfma")):
    m = engine.Model(qpath, quant_mode=qm)
    ctx = engine.Context(m, max_batch=B, max_samples=480000)
    ctx.set_precision(prec)
    p = ctx.default_params(); p.suppress_nst = 1
    ctx.full_batch(pcms, p)
    t0 = time.perf_counter()
    res = ctx.full_batch(pcms, p)
    dt = time.perf_counter() - t0
    t = ctx.timing()
    print("%-40s %8.1f x real time  (step %.1f ms: mel %.1f, encode %.1f, decode %.1f; %d tokens, %d decode steps)"
          % (label + " [%s]" % kind, B * 30.0 / dt, dt * 1e3, t["mel_ms"], t["encode_ms"], t["decode_ms"], sum(len(r["tokens"]) for r in res), t["n_decode_steps"]), flush=True)
    if "--profile" in sys.argv:      # per kernel class, HIP event pairs around every launch (eager, one row group): where the step goes
        ctx.profile(True); ctx.full_batch(pcms, p); prof = ctx.profile_get(); ctx.profile(False)
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            if v["count"]:
                print("    %-18s %6d launches %9.2f ms  (%.1f us each)" % (k, v["count"], v["ms"], 1e3 * v["ms"] / v["count"]), flush=True)
    del ctx; m.close()
