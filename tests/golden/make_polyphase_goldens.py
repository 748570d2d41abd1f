#!/usr/bin/env python3
"""Writes tests/golden/polyphase_upfirdn.json: outputs of scipy.signal.upfirdn (float64) with the documented Kaiser taps (tests/polyphase_ref.py)
on seeded signals, sampled at fixed positions — the committed known answers k_resample_polyphase is held to (<= 1e-5) on the GPU box.

    python tests/golden/make_polyphase_goldens.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import polyphase_ref as pr  # noqa: E402

CASES = [(48000, 16000, 1, 48000, 11), (44100, 16000, 1, 44100, 12), (44100, 16000, 2, 22050, 13), (32000, 16000, 2, 16000, 14), (8000, 16000, 1, 8000, 15), (22050, 16000, 1, 11025, 16)]


def positions(n_out):
    head = list(range(0, min(48, n_out)))
    tail = list(range(max(0, n_out - 48), n_out))
    mid = list(range(48, max(48, n_out - 48), max(1, n_out // 160)))
    return sorted(set(head + mid + tail))


def main():
    import scipy
    out = {"about": "scipy %s signal.upfirdn, float64, taps from tests/polyphase_ref.design; x = polyphase_ref.test_signal(seed, frames, channels, in_rate)" % scipy.__version__, "cases": []}
    for in_rate, out_rate, ch, frames, seed in CASES:
        x = pr.test_signal(seed, frames, ch, in_rate)
        y = pr.reference(x, ch, in_rate, out_rate).reshape(-1, ch)
        pos = positions(y.shape[0])
        out["cases"].append({"in_rate": in_rate, "out_rate": out_rate, "channels": ch, "frames": frames, "seed": seed, "n_out": int(y.shape[0]),
                             "positions": pos, "values": [[float("%.9g" % v) for v in y[p]] for p in pos]})
        print(in_rate, out_rate, ch, "n_out", y.shape[0], "rms", float(np.sqrt(np.mean(y ** 2))))
    with open(os.path.join(HERE, "polyphase_upfirdn.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
