#!/usr/bin/env python3
"""Experiment: do two engine contexts driven from two host threads overlap on one GPU (one's latency-bound decode under the other's
MFMA-bound encode)?  Prints batches/s for 1 and 2 concurrent contexts.  usage: python tools/exp_overlap.py [--clips 64] [--steps 4]"""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import synth_model
from streamkit_amd import engine, synth

ap = argparse.ArgumentParser(); ap.add_argument("--clips", type=int, default=64); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--contexts", type=int, default=2)
a = ap.parse_args()
model = engine.Model(synth_model("small"))
host = np.stack([synth.clip(c, 480000) for c in range(a.clips)]); dev = torch.from_numpy(host).cuda(); torch.cuda.synchronize()
ptrs = [dev[i].data_ptr() for i in range(a.clips)]; ns = [480000] * a.clips
ctxs = [engine.Context(model, max_batch=a.clips, max_samples=480000) for _ in range(a.contexts)]
p = ctxs[0].default_params(); p.suppress_nst = 1
for c in ctxs: c.full_batch(None, p, device_ptrs=ptrs, n_samples=ns)
def run(c, k, lag=0.0):
    time.sleep(lag)
    for _ in range(k): c.full_batch(None, p, device_ptrs=ptrs, n_samples=ns)
t0 = time.perf_counter(); run(ctxs[0], a.steps); t1 = time.perf_counter() - t0
print("1 context : %.1f ms per batch" % (1e3 * t1 / a.steps))
for lag in (0.0, 0.24):
    th = [threading.Thread(target=run, args=(c, a.steps, lag * i)) for i, c in enumerate(ctxs)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t2 = time.perf_counter() - t0
    print("%d contexts, start lag %.0f ms: %.1f ms per batch aggregate (%.2fx)" % (a.contexts, lag * 1e3, 1e3 * t2 / (a.steps * a.contexts), t1 / a.steps / (t2 / (a.steps * a.contexts))))
