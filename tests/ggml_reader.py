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


_BLOCK = {2: 18, 3: 20, 6: 22, 7: 24, 8: 34}


def _dequant_blocks(raw, tt):
    """Independent (numpy) decode of ggml q4_0/q4_1/q5_0/q5_1/q8_0 blocks -> float32 [n]; the arithmetic the loaders state:
    q*d exact in f32, + m one f32 rounding."""
    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, _BLOCK[tt])
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0]
    off = 2
    m = None
    if tt in (3, 7):
        m = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0]; off = 4
    if tt == 8:
        q = b[:, 2:].copy().view(np.int8).astype(np.float32)
        return (q * d[:, None]).astype(np.float32).reshape(-1)
    qh = None
    if tt in (6, 7):
        qh = b[:, off:off + 4].copy().view(np.uint32)[:, 0]; off += 4
    qs = b[:, off:off + 16]
    lo = (qs & 0x0F).astype(np.int32); hi = (qs >> 4).astype(np.int32)
    if qh is not None:
        j = np.arange(16, dtype=np.uint32)
        lo |= (((qh[:, None] >> j) & 1) << 4).astype(np.int32); hi |= (((qh[:, None] >> (j + 16)) & 1) << 4).astype(np.int32)
    q = np.concatenate([lo, hi], axis=1)
    if tt == 2: q = q - 8
    if tt == 6: q = q - 16
    y = (q.astype(np.float32) * d[:, None]).astype(np.float32)
    if m is not None: y = (y + m[:, None]).astype(np.float32)
    return y.reshape(-1)


def dequantize_file_to_f16(src, dst):
    """Rewrite a block-quantised GGML file as its f16 twin (quantised tensors decoded with _dequant_blocks, rounded to f16)."""
    f = open(src, "rb"); o = open(dst, "wb")
    o.write(f.read(4))
    hp = list(struct.unpack("<11i", f.read(44))); hp[10] = 1; o.write(struct.pack("<11i", *hp))
    n_mel, n_fft = struct.unpack("<2i", f.read(8)); o.write(struct.pack("<2i", n_mel, n_fft)); o.write(f.read(4 * n_mel * n_fft))
    nv, = struct.unpack("<i", f.read(4)); o.write(struct.pack("<i", nv))
    for _ in range(nv):
        ln, = struct.unpack("<I", f.read(4)); o.write(struct.pack("<I", ln)); o.write(f.read(ln))
    n = 0
    while True:
        h = f.read(12)
        if len(h) < 12:
            break
        nd, ln, tt = struct.unpack("<3i", h)
        ne = struct.unpack("<%di" % nd, f.read(4 * nd)); name = f.read(ln); cnt = int(np.prod(ne))
        if tt in _BLOCK:
            raw = _dequant_blocks(f.read(cnt // 32 * _BLOCK[tt]), tt).astype(np.float16).tobytes(); tt = 1; n += 1
        else:
            raw = f.read(cnt * (4 if tt == 0 else 2))
        o.write(struct.pack("<3i", nd, ln, tt)); o.write(struct.pack("<%di" % nd, *ne)); o.write(name); o.write(raw)
    o.close()
    return n
