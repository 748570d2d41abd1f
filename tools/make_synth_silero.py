#!/usr/bin/env python3
"""Writes a Silero-VAD-shaped ONNX file with seeded weights (there is no silero_vad.onnx offline).

Structure mirrors the published v5 export as recalled: a top-level `If` on the sample rate whose two sub-graphs hold the 8 kHz
and the 16 kHz networks (same tensor shapes except the STFT basis and the first convolution), weights stored partly as
sub-graph initializers and partly as `Constant` nodes, raw little-endian f32.  The protobuf is encoded by hand (no `onnx`
package in this image).  The 16 kHz weights are engineered, like the synthetic Whisper decoder, so that the gate behaves:
the STFT basis is a real Hann-windowed DFT, the convolutions average non-negative magnitudes, the LSTM's input/candidate
gates open on loud frames and its forget gate stays shut — probability ~0.02 on silence, ~0.99 on tones above ~-30 dBFS —
with a seeded perturbation on every tensor so that all of them matter to the result.

usage: make_synth_silero.py OUT.onnx [--seed N] [--lstm-op]     (--lstm-op stores W/R/B of an ONNX LSTM operator, gate order iofc)
"""
import struct
import sys

import numpy as np


def varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def vi(field, v):
    return varint(field << 3) + varint(v)


def ld(field, payload):
    return varint((field << 3) | 2) + varint(len(payload)) + payload


def tensor(name, arr):
    arr = np.ascontiguousarray(arr, dtype="<f4")
    b = b"".join(vi(1, d) for d in arr.shape) + vi(2, 1) + ld(8, name.encode()) + ld(9, arr.tobytes())
    return b


def int64_tensor(name, vals):
    a = np.asarray(vals, dtype="<i8")
    return b"".join(vi(1, d) for d in a.shape) + vi(2, 7) + ld(8, name.encode()) + ld(9, a.tobytes())


def attr_tensor(name, t):
    return ld(1, name.encode()) + ld(5, t) + vi(20, 4)


def attr_graph(name, g):
    return ld(1, name.encode()) + ld(6, g) + vi(20, 5)


def node(op, inputs, outputs, attrs=(), name=""):
    b = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    if name:
        b += ld(3, name.encode())
    return b + ld(4, op.encode()) + b"".join(ld(5, a) for a in attrs)


def value_info(name):
    return ld(1, name.encode())


def graph(name, nodes, initializers, inputs=(), outputs=()):
    return (b"".join(ld(1, n) for n in nodes) + ld(2, name.encode()) + b"".join(ld(5, t) for t in initializers) +
            b"".join(ld(11, value_info(i)) for i in inputs) + b"".join(ld(12, value_info(o)) for o in outputs))


def weights_16k(seed):
    rng = np.random.default_rng(seed)
    k = np.arange(256)
    hann = 0.5 - 0.5 * np.cos(2 * np.pi * k / 256)
    basis = np.zeros((258, 1, 256), np.float32)
    for b in range(129):
        basis[b, 0] = hann * np.cos(2 * np.pi * b * k / 256)
        basis[129 + b, 0] = -hann * np.sin(2 * np.pi * b * k / 256)
    w = {"stft.forward_basis_buffer": basis}
    for l, (co, ci) in enumerate([(128, 129), (64, 128), (64, 64), (128, 64)]):
        w["encoder.%d.reparam_conv.weight" % l] = (rng.random((co, ci, 3)) * 2.0 / (ci * 3) + rng.normal(0, 0.02 / (ci * 3), (co, ci, 3))).astype(np.float32)
        w["encoder.%d.reparam_conv.bias" % l] = rng.normal(0, 1e-3, co).astype(np.float32)
    gain = 120.0                                                             # conv3 features are ~0.3 x the tone amplitude
    w_ih = rng.random((512, 128)) * 2.0 * gain / 128
    w_ih[128:256] = 0.0                                                      # forget gate ignores the input
    w_ih += rng.normal(0, 0.01, (512, 128))
    w_hh = rng.normal(0, 0.02, (512, 128))
    b_ih = np.concatenate([np.full(128, -3.0), np.full(128, -3.0), np.zeros(128), np.zeros(128)]) + rng.normal(0, 0.05, 512)
    b_hh = rng.normal(0, 0.05, 512)
    w["decoder.rnn.weight_ih"] = w_ih.astype(np.float32)
    w["decoder.rnn.weight_hh"] = w_hh.astype(np.float32)
    w["decoder.rnn.bias_ih"] = b_ih.astype(np.float32)
    w["decoder.rnn.bias_hh"] = b_hh.astype(np.float32)
    w["decoder.decoder.2.weight"] = (0.1 + rng.normal(0, 0.01, (1, 128, 1))).astype(np.float32)
    w["decoder.decoder.2.bias"] = np.array([-4.0], np.float32)
    return w


def weights_8k(seed):
    rng = np.random.default_rng(seed + 1000)
    w = {"stft.forward_basis_buffer": rng.normal(0, 0.1, (130, 1, 128)).astype(np.float32)}
    for l, (co, ci) in enumerate([(128, 65), (64, 128), (64, 64), (128, 64)]):
        w["encoder.%d.reparam_conv.weight" % l] = rng.normal(0, 0.05, (co, ci, 3)).astype(np.float32)
        w["encoder.%d.reparam_conv.bias" % l] = rng.normal(0, 0.05, co).astype(np.float32)
    for n, shape in [("decoder.rnn.weight_ih", (512, 128)), ("decoder.rnn.weight_hh", (512, 128)), ("decoder.rnn.bias_ih", (512,)), ("decoder.rnn.bias_hh", (512,)),
                     ("decoder.decoder.2.weight", (1, 128, 1)), ("decoder.decoder.2.bias", (1,))]:
        w[n] = rng.normal(0, 0.05, shape).astype(np.float32)
    return w


def to_lstm_op(w):
    """LSTMCell tensors (gate blocks i, f, g, o) -> ONNX LSTM operator tensors W, R [1, 4H, H], B [1, 8H] (gate blocks i, o, f, c)."""
    order = [0, 3, 1, 2]
    def blk(a):
        return np.concatenate([a[128 * g:128 * (g + 1)] for g in order])
    out = {k: v for k, v in w.items() if "rnn" not in k}
    out["decoder.rnn.W"] = blk(w["decoder.rnn.weight_ih"])[None]
    out["decoder.rnn.R"] = blk(w["decoder.rnn.weight_hh"])[None]
    out["decoder.rnn.B"] = np.concatenate([blk(w["decoder.rnn.bias_ih"]), blk(w["decoder.rnn.bias_hh"])])[None]
    return out


def branch(prefix, w):
    """Half of the tensors as initializers, half as Constant nodes (both occur in exported graphs); plus shape constants the reader must skip."""
    inits, nodes = [], []
    for i, (name, arr) in enumerate(w.items()):
        full = prefix + name
        if i % 2 == 0:
            inits.append(tensor(full, arr))
        else:
            nodes.append(node("Constant", [], [full], [attr_tensor("value", tensor("", arr))]))
    nodes.append(node("Constant", [], [prefix + "pads"], [attr_tensor("value", int64_tensor("", [0, 0, 0, 64]))]))
    nodes.append(node("Identity", [prefix + "stft.forward_basis_buffer"], [prefix + "out"]))
    return graph(prefix + "graph", nodes, inits, outputs=[prefix + "out"])


def build(seed=1234, lstm_op=False):
    w16, w8 = weights_16k(seed), weights_8k(seed)
    if lstm_op:
        w16, w8 = to_lstm_op(w16), to_lstm_op(w8)
    g8 = branch("If_0_else_branch__Inline_0__", w8)
    g16 = branch("If_0_then_branch__Inline_0__", w16)
    top_nodes = [node("Constant", [], ["sr16k"], [attr_tensor("value", int64_tensor("", [16000]))]),
                 node("Equal", ["sr", "sr16k"], ["is16k"]),
                 node("If", ["is16k"], ["output"], [attr_graph("else_branch", g8), attr_graph("then_branch", g16)])]     # the 8 kHz sub-graph comes first in the file
    g = graph("silero_vad_synth", top_nodes, [], inputs=["input", "state", "sr"], outputs=["output", "stateN"])
    model = vi(1, 8) + ld(2, b"streamkit_amd.make_synth_silero") + ld(7, g) + ld(8, ld(1, b"") + vi(2, 16))
    return model, w16


if __name__ == "__main__":
    out = sys.argv[1]
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1234
    data, _ = build(seed, "--lstm-op" in sys.argv)
    with open(out, "wb") as f:
        f.write(data)
    print("wrote %s (%d bytes)" % (out, len(data)))
