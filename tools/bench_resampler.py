#!/usr/bin/env python3
"""Resampler front end on the GPU: whole 30 s files through skw_resample_linear (48 kHz and 44.1 kHz mono -> 16 kHz, chunk 960) and
skw_resample_polyphase.  Prints wall time per call (H2D + kernels + D2H through the C ABI); run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times quoted in DESIGN.md (7.68 MB of traffic per 30 s of 48 kHz mono).
Set SKW_RESAMPLE_SCAN=1 to force the single-lane index walk for comparison."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streamkit_amd import engine

dsp = engine.Dsp(0)
rng = np.random.default_rng(0)
for in_rate in (48000, 44100):
    n_chunks = in_rate * 30 // 960
    x = (0.3 * rng.standard_normal(n_chunks * 960)).astype(np.float32)
    for it in range(6):
        st = dsp.linear_stream(16000 / in_rate, 960, 1)
        t0 = time.perf_counter(); y = dsp.resample_linear(st, x, n_chunks); dt = time.perf_counter() - t0
    print("linear    %5d -> 16000 Hz, 30 s mono: %7.3f ms per call, %d frames out, index walk: %s" % (in_rate, dt * 1e3, y.size, ("per-chunk parallel, closed-form proposal proven", "per-chunk parallel, second proposal proven (host-walked chunk starts for long calls, binade stepping for short ones)", "single lane (fallback)")[dsp.last_scan_fallback()]))
    for it in range(6):
        t0 = time.perf_counter(); y = dsp.resample_polyphase(x, 1, in_rate, 16000); dt = time.perf_counter() - t0
    print("polyphase %5d -> 16000 Hz, 30 s mono: %7.3f ms per call, %d frames out" % (in_rate, dt * 1e3, y.size))
