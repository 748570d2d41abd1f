"""Differential hunt for the Kokoro node's synthesiser: random texts (words, punctuation, digits, accented and CJK code points, lengths from one symbol to the token cap), random
voices and speeds through libskw_tts.so and through the CPU checker (oracle/skw_kokoro_oracle.cpp): token ids, durations and every tap through the decoder's output bit for bit,
generator stages within the tolerances of tests/test_gpu_kokoro.py.  Usage (GPU box): python tests/hunt/fuzz_kokoro.py [cases] [seed] [size]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401  (the oracle's OpenMP settings: without them its many short parallel regions spin against each other — 80 s per sentence instead of 1)
import kokoro_lib  # noqa: E402
import test_gpu_kokoro as T  # noqa: E402

WORDS = "the a hello world synthesiser test of and with considerably longer sentence commas semicolons colons token count style row frame grows past few hundred frames café naïve 你好 世界 zebra quiz".split()
PUNCT = [".", ",", "!", "?", ";", ":", " - ", "…", " ", "  ", "\n", "(", ")", "\""]

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    size = sys.argv[3] if len(sys.argv) > 3 else "micro"
    d = kokoro_lib.synth_kokoro_dir(size)
    tts = kokoro_lib.Tts(d); orc = kokoro_lib.OracleTts(d); tts.taps(True)
    bad = refused = 0; t0 = time.time()
    for case in range(cases):
        n_words = int(rng.choice([0, 1, 2, 5, 12, 25]))          # (the CPU checker takes minutes on texts at the token cap: tests/test_gpu_kokoro.py covers that edge once)
        parts = []
        for _ in range(n_words):
            w = str(rng.choice(WORDS))
            if rng.random() < 0.15: w = w.upper()
            if rng.random() < 0.1: w = str(int(rng.integers(0, 100000)))
            parts.append(w); parts.append(str(rng.choice(PUNCT)) if rng.random() < 0.4 else " ")
        text = "".join(parts) or str(rng.choice(["a", ".", "你", "?!"]))
        sid = int(rng.integers(0, 103)); speed = float(rng.choice([0.5, 0.8, 1.0, 1.0, 1.25, 2.0]))
        try:
            y, rate = tts.generate(text, sid, speed)
        except RuntimeError as e:
            if kokoro_lib.tokenize(text, d) == [0, 0] and "no symbol of the text" in str(e):      # nothing but the two pad ids: the product's rule (the checker has no such rule)
                refused += 1
                continue
            try:
                orc.synth(text, sid, speed); bad += 1; print("GPU refused what the checker synthesised: %r (%s)" % (text[:60], e), flush=True)
            except Exception:
                refused += 1
            continue
        r = orc.synth(text, sid, speed)
        ok = tts.tokenize(text).tolist() == kokoro_lib.tokenize(text, d) and np.array_equal(tts.tap(0).astype(np.int32), r["dur"]) and y.size == r["y"].size
        for what, name in ((5, "bert"), (6, "d_en"), (7, "t_en"), (1, "f0"), (2, "en"), (3, "dec")):
            ok = ok and np.array_equal(tts.tap(what).view(np.uint32), r[name].ravel().view(np.uint32))
        if ok:
            keep, wraps = T._wrap_mask(tts.tap(8).reshape(-1, 22), r["har"])
            if keep.mean() > 0.2:
                e = max(T._rel(tts.tap(8).reshape(-1, 22)[keep, :11], r["har"][keep, :11]), T._rel(tts.tap(4).reshape(-1, 22)[keep], r["post"][keep]))
                ok = e < T.TOL_GEN and np.isfinite(y).all()
        if not ok:
            bad += 1; print("MISMATCH case %d: %r sid %d speed %.2f" % (case, text[:80], sid, speed), flush=True)
        print("case %d (%d symbols): %d mismatches, %d refused by both, %.0f s" % (case, len(text), bad, refused, time.time() - t0), flush=True)
    print("DONE: %d cases, %d mismatches, %d refused by both sides" % (cases, bad, refused))
    sys.exit(1 if bad else 0)
