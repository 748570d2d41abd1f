#!/usr/bin/env python3
"""Times libskw_tts.so at Kokoro-82M's geometry (seeded weights, tools/make_synth_kokoro.py --size kokoro82m): N calls of ~30 s of speech each, GPU-event time per call.
usage: kokoro_bench.py [--size kokoro82m] [--tokens 300] [--reps 5]      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")
import kokoro_lib  # noqa: E402


def main():
    a = sys.argv[1:]
    size = a[a.index("--size") + 1] if "--size" in a else "kokoro82m"
    n_tok = int(a[a.index("--tokens") + 1]) if "--tokens" in a else 300
    reps = int(a[a.index("--reps") + 1]) if "--reps" in a else 5
    tts = kokoro_lib.Tts(kokoro_lib.synth_kokoro_dir(size))
    ids = np.concatenate([[0], np.random.default_rng(5).integers(1, 60, n_tok), [0]]).astype(np.int32)
    y, _ = tts.generate(None, 50, 1.0, ids=ids)
    speed = float(np.clip((y.size // 600) / 1200.0, 0.3, 3.0))      # aim at ~1200 frames = 30 s
    ms = []
    for _ in range(reps):
        y, _ = tts.generate(None, 50, speed, ids=ids); ms.append(tts.last_ms())
    secs = y.size / 24000.0
    print("%s: %d tokens -> %d frames, %.1f s of audio; GPU ms per call %s; best %.1f ms = %.0fx real time" % (size, ids.size, y.size // 600, secs, ["%.1f" % m for m in ms], min(ms), secs * 1000 / min(ms)))
    tts.close()


if __name__ == "__main__":
    main()
