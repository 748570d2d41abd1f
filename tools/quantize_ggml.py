#!/usr/bin/env python3
"""Re-encode an f16 whisper GGML file with block-quantised 2-D weights (q4_0 q4_1 q5_0 q5_1 q8_0), the way whisper.cpp's
`quantize` tool lays them out: every 2-D tensor whose row length is a multiple of 32 except the positional embeddings; conv
kernels (3-D), biases and LayerNorm parameters are copied.  Used to make test models for the loader's quantised path
(the reference's default model is ggml-base.en-q5_1.bin).   usage: quantize_ggml.py in.bin out.bin q5_1"""
import struct
import sys
import numpy as np

TYPES = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}
FTYPE = {"q4_0": 2, "q4_1": 3, "q5_0": 8, "q5_1": 9, "q8_0": 7}
SKIP = ("encoder.conv1.bias", "encoder.conv2.bias", "encoder.positional_embedding", "decoder.positional_embedding")


def quantize_blocks(x, kind):
    """x: float32 [nb, 32] -> bytes; ggml's reference quantisers (quantize_row_*_ref) in numpy float32."""
    nb = x.shape[0]
    f1 = np.float32
    if kind == "q8_0":
        amax = np.abs(x).max(axis=1)
        d = (amax / f1(127)).astype(np.float32)
        idv = np.where(d != 0, f1(1) / np.where(d != 0, d, f1(1)), f1(0)).astype(np.float32)
        q = np.sign(x * idv[:, None]) * np.floor(np.abs(x * idv[:, None]) + f1(0.5))        # roundf
        out = np.zeros((nb, 34), np.uint8)
        out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(nb, 2)
        out[:, 2:] = q.astype(np.int8).view(np.uint8)
        return out.tobytes()
    five = kind in ("q5_0", "q5_1")
    levels = 31 if five else 15
    if kind in ("q4_0", "q5_0"):
        idx = np.abs(x).argmax(axis=1)
        mx = x[np.arange(nb), idx]
        d = (mx / f1(-(levels + 1) / 2)).astype(np.float32)
        idv = np.where(d != 0, f1(1) / np.where(d != 0, d, f1(1)), f1(0)).astype(np.float32)
        q = np.minimum(levels, (x * idv[:, None] + f1((levels + 1) / 2 + 0.5)).astype(np.int32)).astype(np.uint8)
        head = d.astype(np.float16).view(np.uint8).reshape(nb, 2)
    else:
        mn = x.min(axis=1); mx = x.max(axis=1)
        d = ((mx - mn) / f1(levels)).astype(np.float32)
        idv = np.where(d != 0, f1(1) / np.where(d != 0, d, f1(1)), f1(0)).astype(np.float32)
        q = np.minimum(levels, ((x - mn[:, None]) * idv[:, None] + f1(0.5)).astype(np.int32)).astype(np.uint8)
        head = np.concatenate([d.astype(np.float16).view(np.uint8).reshape(nb, 2), mn.astype(np.float16).view(np.uint8).reshape(nb, 2)], axis=1)
    lo, hi = q[:, :16], q[:, 16:]
    qs = ((lo & 0x0F) | ((hi & 0x0F) << 4)).astype(np.uint8)
    parts = [head]
    if five:
        qh = np.zeros(nb, np.uint32)
        for j in range(16):
            qh |= ((lo[:, j].astype(np.uint32) >> 4) & 1) << j
            qh |= ((hi[:, j].astype(np.uint32) >> 4) & 1) << (j + 16)
        parts.append(qh.view(np.uint8).reshape(nb, 4))
    parts.append(qs)
    return np.concatenate(parts, axis=1).tobytes()


def main(src, dst, kind):
    f = open(src, "rb"); o = open(dst, "wb")
    o.write(f.read(4))
    hp = list(struct.unpack("<11i", f.read(44))); hp[10] = FTYPE[kind] + 2 * 1000      # ftype + GGML_QNT_VERSION (2) * GGML_QNT_VERSION_FACTOR
    o.write(struct.pack("<11i", *hp))
    n_mel, n_fft = struct.unpack("<2i", f.read(8)); o.write(struct.pack("<2i", n_mel, n_fft)); o.write(f.read(4 * n_mel * n_fft))
    nv, = struct.unpack("<i", f.read(4)); o.write(struct.pack("<i", nv))
    for _ in range(nv):
        ln, = struct.unpack("<I", f.read(4)); o.write(struct.pack("<I", ln)); o.write(f.read(ln))
    nq = 0
    while True:
        h = f.read(12)
        if len(h) < 12:
            break
        nd, ln, tt = struct.unpack("<3i", h)
        ne = struct.unpack("<%di" % nd, f.read(4 * nd)); name = f.read(ln)
        n = int(np.prod(ne)); raw = f.read(n * (4 if tt == 0 else 2))
        if nd == 2 and ne[0] % 32 == 0 and name.decode() not in SKIP and name.endswith(b"weight"):
            x = np.frombuffer(raw, dtype=np.float32 if tt == 0 else np.float16).astype(np.float32).reshape(-1, 32)
            raw = quantize_blocks(x, kind); tt = TYPES[kind]; nq += 1
        o.write(struct.pack("<3i", nd, ln, tt)); o.write(struct.pack("<%di" % nd, *ne)); o.write(name); o.write(raw)
    o.close()
    print("%s: %d tensors quantised to %s" % (dst, nq, kind))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
