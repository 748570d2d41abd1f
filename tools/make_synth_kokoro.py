#!/usr/bin/env python3
"""Writes a Kokoro-shaped model directory with seeded weights (there is no kokoro-multi-lang-v1_1 offline).

The directory has the three files the reference's node insists on (kokoro_node.rs:741-758): model.onnx, voices.bin, tokens.txt — plus a small
lexicon-us-en.txt.  model.onnx holds the initializers of the REDUCED network streamkit_amd/csrc/skw_tts.hip evaluates, by
name, raw little-endian f32, as a protobuf encoded by hand (no `onnx` package in this image); it is not an export of Kokoro-82M.
voices.bin is f32 [n_speakers][510][256] like Kokoro's (one 256-float style row per token count: 128 acoustic + 128 prosody).
Weights are scaled so that speech-like numbers come out: ~2.6 frames (65 ms) per symbol, F0 inside 60..400 Hz, waveform amplitude ~0.1.

usage: make_synth_kokoro.py OUT_DIR [--seed N] [--size micro|small] [--speakers N]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_synth_silero import ld, tensor  # noqa: E402  (the hand protobuf encoder)

SIZES = {"micro": dict(d=64, n_te=2, K=50, C=48, n_dec=2, G=32), "small": dict(d=256, n_te=3, K=50, C=128, n_dec=3, G=64)}
N_SYM, U, H, STYLE = 178, 120, 8, 128
SYMBOLS = "$;:,.!?-'\"() " + "abcdefghijklmnopqrstuvwxyz" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "0123456789" + "éèü。！？你好"


def uni(rng, shape, fan_in, gain=1.0):
    s = gain * np.sqrt(3.0 / fan_in)
    return rng.uniform(-s, s, size=shape).astype(np.float32)


def build(seed, size):
    g = SIZES[size]; d, C, G, K = g["d"], g["C"], g["G"], g["K"]
    rng = np.random.default_rng(seed)
    w = {}
    w["text_encoder.embedding.weight"] = rng.standard_normal((N_SYM, d)).astype(np.float32)
    for i in range(g["n_te"]):
        p = "text_encoder.cnn.%d." % i
        w[p + "weight"] = uni(rng, (d, d, 5), 5 * d, 1.4); w[p + "bias"] = uni(rng, (d,), 16)
        w[p + "norm.gamma"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32); w[p + "norm.beta"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
    w["predictor.text_encoder.fc.weight"] = uni(rng, (2 * d, STYLE), STYLE, 0.5); w["predictor.text_encoder.fc.bias"] = np.zeros(2 * d, np.float32)
    w["predictor.duration_proj.weight"] = uni(rng, (K, d), d, 1.0); w["predictor.duration_proj.bias"] = np.full(K, -3.2, np.float32)
    w["predictor.F0_proj.weight"] = uni(rng, (d,), d, 1.5); w["predictor.F0_proj.style"] = uni(rng, (STYLE,), STYLE, 1.0); w["predictor.F0_proj.bias"] = np.array([-0.6], np.float32)
    w["predictor.N_proj.weight"] = uni(rng, (d,), d, 1.0); w["predictor.N_proj.bias"] = np.array([0.1], np.float32)
    w["decoder.encode.weight"] = uni(rng, (C, d + 2, 3), 3 * (d + 2), 1.4); w["decoder.encode.bias"] = uni(rng, (C,), 16)
    w["decoder.encode.fc.weight"] = uni(rng, (2 * C, STYLE), STYLE, 0.5); w["decoder.encode.fc.bias"] = np.zeros(2 * C, np.float32)
    for i in range(g["n_dec"]):
        p = "decoder.decode.%d." % i
        w[p + "weight"] = uni(rng, (C, C, 3), 3 * C, 1.4); w[p + "bias"] = uni(rng, (C,), 16)
        w[p + "fc.weight"] = uni(rng, (2 * C, STYLE), STYLE, 0.5); w[p + "fc.bias"] = np.zeros(2 * C, np.float32)
    # generator: smooth up-sampling kernels (a raised cosine across the 120 sub-frames times a random channel mix) so the spectrum moves slowly inside a frame
    mix = uni(rng, (C, G), C, 1.0)
    ramp = (0.75 + 0.25 * np.cos(2 * np.pi * (np.arange(U) / U))).astype(np.float32)
    w["decoder.generator.ups.weight"] = (mix[:, :, None] * ramp[None, None, :] + 0.02 * rng.standard_normal((C, G, U))).astype(np.float32)
    w["decoder.generator.ups.bias"] = uni(rng, (G,), 16)
    w["decoder.generator.source.weight"] = (uni(rng, (H, G), H, 1.0) / (1.0 + np.arange(H))[:, None]).astype(np.float32)
    w["decoder.generator.resblock.alpha"] = (1.0 + 0.5 * rng.random(G)).astype(np.float32)
    w["decoder.generator.resblock.weight"] = uni(rng, (G, G, 3), 3 * G, 1.0); w["decoder.generator.resblock.bias"] = np.zeros(G, np.float32)
    post = uni(rng, (22, G, 7), 7 * G, 0.6)
    w["decoder.generator.conv_post.weight"] = post
    pb = np.zeros(22, np.float32); pb[:11] = -0.8 - 0.25 * np.arange(11)      # log-magnitudes falling with frequency
    w["decoder.generator.conv_post.bias"] = pb
    return w


def write_dir(out, seed=1234, size="micro", speakers=103):
    os.makedirs(out, exist_ok=True)
    w = build(seed, size)
    graph = b"".join(ld(5, tensor(k, v)) for k, v in w.items()) + ld(2, b"skw_kokoro_synth")
    open(os.path.join(out, "model.onnx"), "wb").write(ld(7, graph))
    rng = np.random.default_rng(seed + 1)
    base = rng.standard_normal((speakers, 1, 2 * STYLE)).astype(np.float32)
    rows = np.arange(510, dtype=np.float32)[None, :, None] / 510.0
    drift = rng.standard_normal((speakers, 1, 2 * STYLE)).astype(np.float32) * 0.2
    (base + drift * rows).astype("<f4").tofile(os.path.join(out, "voices.bin"))
    with open(os.path.join(out, "tokens.txt"), "w", encoding="utf-8") as f:
        for i, ch in enumerate(SYMBOLS):
            f.write("%s %d\n" % (ch, i))
    with open(os.path.join(out, "lexicon-us-en.txt"), "w", encoding="utf-8") as f:
        f.write("hello h e l o\nworld w r l d\nthe d a\nthe t h e\n")


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        sys.exit(__doc__)
    seed = int(a[a.index("--seed") + 1]) if "--seed" in a else 1234
    size = a[a.index("--size") + 1] if "--size" in a else "micro"
    spk = int(a[a.index("--speakers") + 1]) if "--speakers" in a else 103
    write_dir(a[0], seed, size, spk)
