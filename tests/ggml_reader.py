"""Minimal reader of whisper.cpp's legacy GGML container (test-side, numpy)."""
import struct
import numpy as np


def read_ggml(path):
    f = open(path, "rb")
    magic, = struct.unpack("<i", f.read(4))
    assert magic == 0x67676d6c
    names = ("n_vocab", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_mels", "ftype")
    hp = dict(zip(names, struct.unpack("<11i", f.read(44))))
    n_mel, n_fft = struct.unpack("<2i", f.read(8))
    filters = np.frombuffer(f.read(4 * n_mel * n_fft), dtype=np.float32).reshape(n_mel, n_fft).copy()
    nv, = struct.unpack("<i", f.read(4))
    vocab = []
    for _ in range(nv):
        ln, = struct.unpack("<I", f.read(4))
        vocab.append(f.read(ln))
    tensors = {}
    while True:
        h = f.read(12)
        if len(h) < 12:
            break
        nd, ln, tt = struct.unpack("<3i", h)
        ne = struct.unpack("<%di" % nd, f.read(4 * nd))
        name = f.read(ln).decode()
        n = int(np.prod(ne))
        dt = np.float32 if tt == 0 else np.float16
        data = np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt).reshape(tuple(reversed(ne))).copy()
        tensors[name] = data
    return hp, filters, vocab, tensors
