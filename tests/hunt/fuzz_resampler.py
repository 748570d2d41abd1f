"""Differential hunt for the linear resampler (R1): random rate pairs (the audio world's rates and odd ones), chunk sizes, channel counts and call sizes through the GPU path
(skw_dsp, streaming state carried across calls) and through the oracle's rubato restatement, chunk by chunk: every output sample bit-identical, every output length equal.
Usage (GPU box): python tests/hunt/fuzz_resampler.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from streamkit_amd import engine  # noqa: E402

RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 192000, 7999, 16001, 44056, 47952]

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dsp = engine.Dsp(0); bad = 0; t0 = time.time(); how = {0: 0, 1: 0, 2: 0}
    for case in range(cases):
        fin, fout = int(rng.choice(RATES)), int(rng.choice(RATES))
        if fin == fout:
            fout = 16000 if fin != 16000 else 48000
        ch = int(rng.integers(1, 3)); chunk = int(rng.choice([64, 128, 160, 441, 480, 512, 960, 1024, 1920, 4096])); n_chunks = int(rng.integers(1, 60))
        x = (rng.standard_normal(chunk * n_chunks * ch) * rng.choice([1e-3, 0.3, 1.0])).astype(np.float32)
        st = dsp.linear_stream(fout / fin, chunk, ch); outs = []; pos = 0
        while pos < n_chunks:
            k = min(n_chunks - pos, int(rng.choice([1, 1, 2, 3, 8, 17, 60])))
            outs.append(dsp.resample_linear(st, x[pos * chunk * ch:(pos + k) * chunk * ch], k)); how[dsp.last_scan_fallback()] += 1; pos += k
        got = np.concatenate(outs).reshape(-1, ch)
        orc = oracle_lib.OracleResampler(fout / fin, chunk, ch)
        ref = np.concatenate([orc.process(x[c * chunk * ch:(c + 1) * chunk * ch].reshape(chunk, ch).T).T for c in range(n_chunks)])
        ok = got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        if not ok:
            bad += 1
            print("MISMATCH case %d: %d -> %d Hz, %d ch, chunk %d x %d: shapes %s / %s, differing samples %s" % (case, fin, fout, ch, chunk, n_chunks, got.shape, ref.shape,
                  int((got.view(np.uint32) != ref.view(np.uint32)).sum()) if got.shape == ref.shape else "-"), flush=True)
        if case % 20 == 19:
            print("case %d: %d mismatches, index walks by path (first proposal / second / single-lane) %s, %.0f s" % (case, bad, how, time.time() - t0), flush=True)
    print("DONE: %d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)
