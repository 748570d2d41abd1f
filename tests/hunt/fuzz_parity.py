"""Differential hunt: random clips (lengths 0.05 - 65 s, levels from silence to clipping, random batch compositions) and random decode parameters through the engine and through the
oracle, tiny model, exact precision: every transcript must be identical (ids, log-probs, segment times, window counts).  With a third argument `f16`: the same inputs through
streamkit_amd.parity.teacher_forced_compare instead — f16_mfma fed the exact precision's tokens, every decision equal or at a near-tie, logits within the precision's bound.
Usage (GPU box): python tests/hunt/fuzz_parity.py [rounds] [seed] [f16 | exact [size [quant kind | all]]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model  # noqa: E402
from oracle_lib import OracleModel  # noqa: E402
from streamkit_amd import engine, synth  # noqa: E402


def same(rg, ro):
    return ([t[0] for t in rg["tokens"]] == [t[0] for t in ro["tokens"]] and [t[3] for t in rg["tokens"]] == [t[3] for t in ro["tokens"]] and
            [(s["t0"], s["t1"], s["text"]) for s in rg["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in ro["segments"]] and
            rg["n_windows"] == ro["n_windows"] and rg["fallback_requested"] == ro["fallback_requested"])


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    f16 = len(sys.argv) > 3 and sys.argv[3] == "f16"      # (third argument: `f16` or `exact`)
    variants = [dict(), dict(vocab=51864), dict(vocab=51866, mels=128)]
    bad = n = 0; t0 = time.time()
    for r in range(rounds):
        v = variants[r % 3]
        size = "tiny" if r % 3 == 0 else "micro"
        if len(sys.argv) > 4:                  # a size named: that geometry only (f16 mode needs no CPU run; `exact small` costs ~6 s of 16 host threads per 30 s clip)
            size, v = sys.argv[4], dict()
        path = synth_model(size, **v)
        if len(sys.argv) > 5:                  # a block-quantised re-encoding of the file (ggml's arithmetic in the exact precision): q4_0 | q4_1 | q5_0 | q5_1 | q8_0 | all (in turn)
            from conftest import quantized_model
            kinds = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"]
            path = quantized_model(size, kinds[r % 5] if sys.argv[5] == "all" else sys.argv[5], **v)
        m = engine.Model(path); ctx = engine.Context(m, max_batch=6, max_samples=16000 * 66); om = OracleModel(path)
        nb = int(rng.integers(1, 7)); clips = []
        for i in range(nb):
            kind = rng.integers(0, 6)
            secs = [0.05, 0.4, 1.5, float(rng.uniform(2, 29)), 30.0 + float(rng.uniform(0, 0.1)), float(rng.uniform(30, 65))][kind]
            x = synth.clip(int(rng.integers(0, 10000)), max(1, int(secs * 16000)))
            g = [0.0, 1e-4, 0.3, 1.0, 4.0][int(rng.integers(0, 5))]
            x = np.clip(x * np.float32(g), -1.0, 1.0).astype(np.float32) if g != 1.0 else x
            clips.append(x)
        p = ctx.default_params(); po = om.default_params()
        for f, choices in [("suppress_blank", [0, 1]), ("suppress_nst", [0, 1]), ("temperature_inc", [0.0, 0.2]), ("no_timestamps", [0, 0, 0, 1]), ("single_segment", [0, 0, 0, 1]),
                           ("max_initial_ts", [1.0, 1.0, 0.3]), ("entropy_thold", [2.4, 2.4, 3.2]), ("logprob_thold", [-1.0, -1.0, -0.3])]:
            val = choices[int(rng.integers(0, len(choices)))]
            setattr(p, f, type(getattr(p, f))(val)); setattr(po, f, type(getattr(po, f))(val))
        if f16:
            from streamkit_amd.parity import teacher_forced_compare
            tf = teacher_forced_compare(ctx, clips, params=p); n += len(clips)
            if not tf["ok"]:
                bad += 1
                print("F16 BOUND round %d model %s: %s" % (r, os.path.basename(path), {k: tf[k] for k in ("steps_checked", "argmax_disagreements", "max_margin_at_disagreement", "max_logit_err")}), flush=True)
            print("round %d: %d clips so far, %d rounds out of bounds; this round %d decisions, %d differ, max logit error %.3f, %.0f s" % (r, n, bad, tf["steps_checked"], tf["argmax_disagreements"],
                  tf["max_logit_err"], time.time() - t0), flush=True)
            ctx.close(); m.close(); om.close()
            continue
        res = ctx.full_batch(clips, params=p)
        for i, (x, rg) in enumerate(zip(clips, res)):
            ro = om.full(x, po); n += 1
            if not same(rg, ro):
                bad += 1
                print("MISMATCH round %d clip %d (%d samples) model %s params %s\n  gpu %s\n  cpu %s" % (r, i, x.size, os.path.basename(path), {f: getattr(p, f) for f, _ in p._fields_},
                      [t[0] for t in rg["tokens"]][:16], [t[0] for t in ro["tokens"]][:16]), flush=True)
        ctx.close(); m.close(); om.close()
        print("round %d: %d clips so far, %d mismatches, %.0f s" % (r, n, bad, time.time() - t0), flush=True)
    print("DONE: %d clips, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)
