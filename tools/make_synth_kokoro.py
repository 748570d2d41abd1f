#!/usr/bin/env python3
"""Writes a Kokoro model directory with seeded weights of the PUBLISHED architecture (there is no kokoro-multi-lang-v1_1 offline).

The directory has the three files the reference's node insists on (kokoro_node.rs:741-758): model.onnx, voices.bin, tokens.txt — plus a small
lexicon-us-en.txt.  model.onnx holds, as ONNX initializers (raw little-endian f32, protobuf encoded by hand: no `onnx` package in this image), every tensor
of the network include/skw_kokoro_net.h wires up, under the PyTorch module names listed there (weight-norm pairs folded into `.weight`):
ALBERT text encoder (one shared layer), bert_encoder, the prosody predictor's BiLSTMs / AdaLayerNorms / AdainResBlk1d stacks, the acoustic text encoder,
the AdaIN decoder and the ISTFTNet generator.  Sizes: "kokoro82m" is Kokoro-82M's geometry (81.8 M parameters — for timing on the GPU box only);
"micro" and "small" are the same network at reduced widths for the tests.  It is not an export of the trained model.
voices.bin is f32 [n_speakers][510][256] like Kokoro's (one 256-float style row per token count: 128 acoustic + 128 prosody).
Weights are scaled so that speech-like numbers come out: ~2 frames per symbol, F0 around 100 - 250 Hz with unvoiced stretches, waveform amplitude ~0.1.

usage: make_synth_kokoro.py OUT_DIR [--seed N] [--size micro|small|kokoro82m] [--speakers N]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_synth_silero import ld, tensor  # noqa: E402  (the hand protobuf encoder)

SIZES = {
    "micro": dict(emb=32, hid=64, ffn=128, layers=2, d=64, max_dur=50, te_depth=2, dec_c=96, asr_c=16, gen=(64, 32, 16), n_decode=2),
    "small": dict(emb=64, hid=128, ffn=256, layers=3, d=128, max_dur=50, te_depth=3, dec_c=192, asr_c=32, gen=(128, 64, 32), n_decode=3),
    "kokoro82m": dict(emb=128, hid=768, ffn=2048, layers=12, d=512, max_dur=50, te_depth=3, dec_c=1024, asr_c=64, gen=(512, 256, 128), n_decode=4),
}
N_SYM, STYLE, MAX_POS = 178, 128, 512
SYMBOLS = "$;:,.!?-'\"() " + "abcdefghijklmnopqrstuvwxyz" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "0123456789" + "éèü。！？你好"


def uni(rng, shape, fan_in, gain=1.0):
    s = gain * np.sqrt(3.0 / fan_in)
    return rng.uniform(-s, s, size=shape).astype(np.float32)


def build(seed, size):
    g = SIZES[size]
    emb, hid, ffn, d, H = g["emb"], g["hid"], g["ffn"], g["d"], g["d"] // 2
    c0, c1, c2 = g["gen"]
    rng = np.random.default_rng(seed)
    w = {}

    def lin(name, n_out, n_in, gain=1.0, bias=True):
        w[name + ".weight"] = uni(rng, (n_out, n_in), n_in, gain)
        if bias:
            w[name + ".bias"] = uni(rng, (n_out,), 64, 0.5)

    def conv(name, c_out, c_in, k, gain=1.0, bias=True):
        w[name + ".weight"] = uni(rng, (c_out, c_in, k), c_in * k, gain)
        if bias:
            w[name + ".bias"] = uni(rng, (c_out,), 64, 0.5)

    def lstm(p, n_in):
        for sfx in ("", "_reverse"):
            w[p + "weight_ih_l0" + sfx] = uni(rng, (4 * H, n_in), n_in, 1.0)
            w[p + "weight_hh_l0" + sfx] = uni(rng, (4 * H, H), H, 0.8)
            w[p + "bias_ih_l0" + sfx] = uni(rng, (4 * H,), 64, 0.5)
            w[p + "bias_hh_l0" + sfx] = uni(rng, (4 * H,), 64, 0.5)

    def style_fc(name, c):
        w[name + ".weight"] = uni(rng, (2 * c, STYLE), STYLE, 0.4)
        w[name + ".bias"] = np.zeros(2 * c, np.float32)

    def resblk(p, c_in, c_out, up):
        style_fc(p + "norm1.fc", c_in); style_fc(p + "norm2.fc", c_out)
        conv(p + "conv1", c_out, c_in, 3, 1.4); conv(p + "conv2", c_out, c_out, 3, 1.4)
        if c_in != c_out:
            conv(p + "conv1x1", c_out, c_in, 1, 1.0, bias=False)
        if up:
            w[p + "pool.weight"] = (np.array([0.5, 1.0, 0.5], np.float32)[None, None, :] * (1.0 + 0.1 * rng.standard_normal((c_in, 1, 1)))).astype(np.float32)
            w[p + "pool.bias"] = np.zeros(c_in, np.float32)

    def resblock1(p, c, k):
        for j in range(3):
            conv(p + "convs1.%d" % j, c, c, k, 0.9); conv(p + "convs2.%d" % j, c, c, k, 0.9)
            style_fc(p + "adain1.%d.fc" % j, c); style_fc(p + "adain2.%d.fc" % j, c)
            w[p + "alpha1.%d" % j] = (1.0 + 0.5 * rng.random(c)).astype(np.float32)
            w[p + "alpha2.%d" % j] = (1.0 + 0.5 * rng.random(c)).astype(np.float32)

    # ---- bert (ALBERT) ----
    w["bert.embeddings.word_embeddings.weight"] = rng.standard_normal((N_SYM, emb)).astype(np.float32)
    w["bert.embeddings.position_embeddings.weight"] = (0.3 * rng.standard_normal((MAX_POS, emb))).astype(np.float32)
    w["bert.embeddings.token_type_embeddings.weight"] = (0.1 * rng.standard_normal((2, emb))).astype(np.float32)
    w["bert.embeddings.LayerNorm.weight"] = (1.0 + 0.1 * rng.standard_normal(emb)).astype(np.float32)
    w["bert.embeddings.LayerNorm.bias"] = (0.1 * rng.standard_normal(emb)).astype(np.float32)
    lin("bert.encoder.embedding_hidden_mapping_in", hid, emb)
    L = "bert.encoder.albert_layer_groups.0.albert_layers.0."
    for p in ("attention.query", "attention.key", "attention.value", "attention.dense"):
        lin(L + p, hid, hid, 1.2 if p != "attention.dense" else 0.7)
    for p in ("attention.LayerNorm", "full_layer_layer_norm"):
        w[L + p + ".weight"] = (1.0 + 0.1 * rng.standard_normal(hid)).astype(np.float32)
        w[L + p + ".bias"] = (0.1 * rng.standard_normal(hid)).astype(np.float32)
    w[L + "ffn.weight"] = uni(rng, (ffn, hid), hid, 1.0); w[L + "ffn.bias"] = uni(rng, (ffn,), 64, 0.5)
    w[L + "ffn_output.weight"] = uni(rng, (hid, ffn), ffn, 0.7); w[L + "ffn_output.bias"] = uni(rng, (hid,), 64, 0.5)
    w["bert.config.num_hidden_layers"] = np.array([g["layers"]], np.float32)
    lin("bert_encoder", d, hid)
    # ---- predictor ----
    for i in range(3):
        lstm("predictor.text_encoder.lstms.%d." % (2 * i), d + STYLE)
        style_fc("predictor.text_encoder.lstms.%d.fc" % (2 * i + 1), d)
    lstm("predictor.lstm.", d + STYLE)
    lstm("predictor.shared.", d + STYLE)
    # the 50 duration bins share a direction, so that a token's bins move together and durations spread over ~1 - 6 frames
    w["predictor.duration_proj.linear_layer.weight"] = (uni(rng, (1, d), d, 4.0) + uni(rng, (g["max_dur"], d), d, 1.0)).astype(np.float32)
    w["predictor.duration_proj.linear_layer.bias"] = np.full(g["max_dur"], -3.3, np.float32)
    for br in ("predictor.F0.", "predictor.N."):
        resblk(br + "0.", d, d, False); resblk(br + "1.", d, d // 2, True); resblk(br + "2.", d // 2, d // 2, False)
    w["predictor.F0_proj.weight"] = uni(rng, (1, d // 2, 1), d // 2, 100.0); w["predictor.F0_proj.bias"] = np.array([110.0], np.float32)      # Hz; dips below 10 Hz read as unvoiced
    w["predictor.N_proj.weight"] = uni(rng, (1, d // 2, 1), d // 2, 1.0); w["predictor.N_proj.bias"] = np.array([0.2], np.float32)
    # ---- text encoder ----
    w["text_encoder.embedding.weight"] = rng.standard_normal((N_SYM, d)).astype(np.float32)
    for i in range(g["te_depth"]):
        conv("text_encoder.cnn.%d.0" % i, d, d, 5, 1.4)
        w["text_encoder.cnn.%d.1.gamma" % i] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
        w["text_encoder.cnn.%d.1.beta" % i] = (0.1 * rng.standard_normal(d)).astype(np.float32)
    lstm("text_encoder.lstm.", d)
    # ---- decoder ----
    conv("decoder.asr_res.0", g["asr_c"], d, 1)
    w["decoder.F0_conv.weight"] = np.array([[[0.25, 0.5, 0.25]]], np.float32) / 200.0; w["decoder.F0_conv.bias"] = np.array([-0.6], np.float32)
    w["decoder.N_conv.weight"] = np.array([[[0.25, 0.5, 0.25]]], np.float32); w["decoder.N_conv.bias"] = np.array([0.0], np.float32)
    resblk("decoder.encode.", d + 2, g["dec_c"], False)
    cat = g["dec_c"] + 2 + g["asr_c"]
    for i in range(g["n_decode"]):
        last = i + 1 == g["n_decode"]
        resblk("decoder.decode.%d." % i, cat, c0 if last else g["dec_c"], last)
    # ---- generator ----
    for i, (ci, co, up) in enumerate(((c0, c1, 10), (c1, c2, 6))):
        ramp = (0.5 - 0.5 * np.cos(2 * np.pi * (np.arange(2 * up) + 0.5) / (2 * up))).astype(np.float32)      # overlapping Hann ramps: smooth up-sampling
        w["decoder.generator.ups.%d.weight" % i] = (uni(rng, (ci, co, 1), ci, 1.2) * ramp[None, None, :] + 0.02 * rng.standard_normal((ci, co, 2 * up))).astype(np.float32)
        w["decoder.generator.ups.%d.bias" % i] = uni(rng, (co,), 64, 0.5)
    w["decoder.generator.m_source.l_linear.weight"] = (uni(rng, (1, 9), 9, 6.0) / (1.0 + np.arange(9))[None, :]).astype(np.float32)
    w["decoder.generator.m_source.l_linear.bias"] = np.array([0.0], np.float32)
    conv("decoder.generator.noise_convs.0", c1, 22, 12, 0.6); conv("decoder.generator.noise_convs.1", c2, 22, 1, 0.6)
    resblock1("decoder.generator.noise_res.0.", c1, 7); resblock1("decoder.generator.noise_res.1.", c2, 11)
    for i in range(2):
        for j, k in enumerate((3, 7, 11)):
            resblock1("decoder.generator.resblocks.%d." % (3 * i + j), c2 if i else c1, k)
    w["decoder.generator.conv_post.weight"] = uni(rng, (22, c2, 7), 7 * c2, 0.5)
    pb = np.zeros(22, np.float32); pb[:11] = -2.2 - 0.25 * np.arange(11)      # log-magnitudes falling with frequency
    w["decoder.generator.conv_post.bias"] = pb
    return w


def write_dir(out, seed=1234, size="micro", speakers=103):
    os.makedirs(out, exist_ok=True)
    w = build(seed, size)
    graph = b"".join(ld(5, tensor(k, v)) for k, v in w.items()) + ld(2, b"skw_kokoro_synth")
    open(os.path.join(out, "model.onnx"), "wb").write(ld(7, graph))
    rng = np.random.default_rng(seed + 1)
    base = (0.3 * rng.standard_normal((speakers, 1, 2 * STYLE))).astype(np.float32)      # (Kokoro's style rows are small numbers too)
    rows = np.arange(510, dtype=np.float32)[None, :, None] / 510.0
    drift = rng.standard_normal((speakers, 1, 2 * STYLE)).astype(np.float32) * 0.06
    (base + drift * rows).astype("<f4").tofile(os.path.join(out, "voices.bin"))
    with open(os.path.join(out, "tokens.txt"), "w", encoding="utf-8") as f:
        for i, ch in enumerate(SYMBOLS):
            f.write("%s %d\n" % (ch, i))
    with open(os.path.join(out, "lexicon-us-en.txt"), "w", encoding="utf-8") as f:
        f.write("hello h e l o\nworld w r l d\nthe d a\nthe t h e\n")
    return sum(v.size for v in w.values())


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        sys.exit(__doc__)
    seed = int(a[a.index("--seed") + 1]) if "--seed" in a else 1234
    size = a[a.index("--size") + 1] if "--size" in a else "micro"
    spk = int(a[a.index("--speakers") + 1]) if "--speakers" in a else 103
    n = write_dir(a[0], seed, size, spk)
    print("%s: %.1f M parameters" % (size, n / 1e6))
