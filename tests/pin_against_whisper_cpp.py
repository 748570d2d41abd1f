#!/usr/bin/env python3
"""Pin this build against the REAL reference, for whoever has what the build environment never had: a trained GGML Whisper file, a 16 kHz mono WAV, and whisper.cpp's own output
for that pair.  (Every parity claim in this repository is "oracle-exact, parity unpinned": the oracle restates whisper.cpp from memory and no reference output existed offline.)

  1. with the reference's whisper.cpp (the revision whisper-rs-sys 0.14.1 bundles, or any v1.7.x):
         whisper-cli -m ggml-base.en.bin -f clip.wav -l en -bs 1 -bo 1 -nf -ojf -of clip          (greedy, no temperature fallback, full JSON with token ids -> clip.json)
     (with the fallback ladder left on — the node's setting — drop `-nf`; a clip that needs a temperature > 0 then depends on whisper.cpp's std::mt19937 stream, which the oracle
      restates too; start with -nf.)
  2. python tests/pin_against_whisper_cpp.py --model ggml-base.en.bin --wav clip.wav --json clip.json [--gpu] [--language en]

What is compared, segment by segment: token ids (the north star's "token ids bit-exact (greedy)"), segment start / end (10 ms units), and the text.  The first divergence is printed
with both sides' tokens around it and the oracle's top1 - top2 margin there: a divergence at a large margin is a restatement error to fix in oracle/ (and then in the kernels, which
are held to the oracle bit for bit); one at a tiny margin is a summation-order effect of ggml's SIMD dot products, which depend on the CPU the reference ran on.
--gpu also runs the HIP engine (exact precision) and reports whether it equals the oracle on this clip (it must)."""
import argparse
import json
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def read_wav_16k_mono(path):
    with wave.open(path, "rb") as w:
        if w.getframerate() != 16000 or w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise SystemExit("%s: need 16 kHz mono PCM16 (whisper-cli's own input format); got %d Hz, %d ch, %d-byte samples" % (path, w.getframerate(), w.getnchannels(), w.getsampwidth()))
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return (pcm.astype(np.float32) / 32768.0).astype(np.float32)      # whisper.cpp's read_wav: int16 / 32768.0f


def reference_segments(doc):
    """whisper-cli -ojf: {"transcription": [{"offsets": {"from": ms, "to": ms}, "text": "...", "tokens": [{"id": n, "text": "...", ...}, ...]}, ...]} -> [(t0, t1, [ids] | None, text)]"""
    out = []
    for seg in doc.get("transcription", []):
        off = seg.get("offsets", {})
        ids = [int(t["id"]) for t in seg["tokens"]] if "tokens" in seg else None
        out.append((int(off.get("from", 0)) // 10, int(off.get("to", 0)) // 10, ids, seg.get("text", "")))
    return out


def compare(ref, ours, eot, margins=None):
    """-> (n_segments_equal, first divergence description or None).  Token lists are compared on ids below <|endoftext|> plus timestamp tokens, as both sides store them."""
    n_ok = 0
    for i, r in enumerate(ref):
        if i >= len(ours):
            return n_ok, "the reference has %d segments, this build %d; first missing: %r" % (len(ref), len(ours), r[3][:80])
        o = ours[i]
        if (r[0], r[1]) != (o[0], o[1]):
            return n_ok, "segment %d: times differ — reference [%d, %d], this build [%d, %d] (10 ms units); texts %r / %r" % (i, r[0], r[1], o[0], o[1], r[3][:60], o[3][:60])
        if r[2] is not None:
            a = [t for t in r[2] if t != eot]; b = [t for t in o[2] if t != eot]
            if a != b:
                k = next((j for j, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                m = "" if margins is None or i >= len(margins) or k >= len(margins[i]) else "; the oracle's top1 - top2 margin at that decision: %.4f" % margins[i][k]
                return n_ok, "segment %d, token %d: reference %s, this build %s%s" % (i, k, a[max(0, k - 3):k + 3], b[max(0, k - 3):k + 3], m)
        elif r[3].strip() != o[3].strip():
            return n_ok, "segment %d: text differs — %r / %r (the JSON holds no token ids: run whisper-cli with -ojf)" % (i, r[3][:80], o[3][:80])
        n_ok += 1
    if len(ours) > len(ref):
        return n_ok, "this build has %d segments, the reference %d; first extra: %r" % (len(ours), len(ref), ours[len(ref)][3][:80])
    return n_ok, None


def segments_of(result):
    """oracle / engine result dict -> [(t0, t1, [ids], text)] and per-segment margins"""
    segs, margins = [], []
    toks = result["tokens"]; pos = 0
    for s in result["segments"]:
        ids = s.get("tokens")
        if ids is None:                       # the oracle's dict keeps tokens flat: take the next len() of them
            n = s.get("n_tokens", 0); ids = [t[0] for t in toks[pos:pos + n]]
        m = [t[4] for t in toks[pos:pos + len(ids)]] if toks and len(toks[0]) > 4 else []
        pos += len(ids)
        segs.append((int(s["t0"]), int(s["t1"]), list(ids), s["text"].decode("utf-8", "replace") if isinstance(s["text"], bytes) else s["text"])); margins.append(m)
    return segs, margins


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", required=True); ap.add_argument("--wav", required=True); ap.add_argument("--json", required=True)
    ap.add_argument("--language", default="en"); ap.add_argument("--no-fallback", action="store_true", default=True, help="temperature_inc = 0 (whisper-cli -nf); pass --fallback for the node's ladder")
    ap.add_argument("--fallback", dest="no_fallback", action="store_false"); ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args(argv)
    from oracle_lib import OracleModel
    pcm = read_wav_16k_mono(a.wav); ref = reference_segments(json.load(open(a.json, encoding="utf-8")))
    om = OracleModel(a.model); po = om.default_params(); po.suppress_nst = 0
    if a.no_fallback:
        po.temperature_inc = 0.0
    from streamkit_amd import engine as _e
    lang = -1 if a.language == "auto" else _e.lib().skw_model_lang_id(a.language.encode())      # whisper.cpp's language table (codes and full names); needs no GPU
    if lang < 0 and a.language != "auto":
        raise SystemExit("unknown language %r" % a.language)
    po.lang_id = lang
    ro = om.full(pcm, po); ours, margins = segments_of(ro)
    eot = om.hp.n_vocab >= 51865 and 50257 or 50256
    n_ok, why = compare(ref, ours, eot, margins)
    print("oracle vs whisper.cpp: %d of %d segments identical%s" % (n_ok, len(ref), "" if why is None else "\n  FIRST DIVERGENCE: " + why))
    rc = 0 if why is None else 1
    if a.gpu:
        from streamkit_amd import engine
        m = engine.Model(a.model); ctx = engine.Context(m, max_batch=1, max_samples=max(pcm.size, 16000)); p = ctx.default_params(); p.lang_id = lang
        if a.no_fallback:
            p.temperature_inc = 0.0
        rg = ctx.full_batch([pcm], params=p)[0]
        same = [t[0] for t in rg["tokens"]] == [t[0] for t in ro["tokens"]] and [(s["t0"], s["t1"]) for s in rg["segments"]] == [(s["t0"], s["t1"]) for s in ro["segments"]]
        print("engine (exact) vs oracle on this clip: %s" % ("identical" if same else "DIFFERENT — a defect of this build, independent of the reference: please report with the clip"))
        rc = rc or (0 if same else 2)
    return rc


if __name__ == "__main__":
    sys.exit(main())
