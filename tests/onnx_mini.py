"""A third, independent reader of the ONNX subset (for the tests): returns every float tensor with the id of the (sub)graph it
was found in, so the tests can bind the 16 kHz Silero weights themselves and hand them to the oracle."""
import struct

import numpy as np


def _varint(b, i):
    r = 0; sh = 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7F) << sh
        if not c & 0x80:
            return r, i
        sh += 7


def _fields(b):
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            n, i = _varint(b, i); v = b[i:i + n]; i += n
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield num, wt, v


def _tensor(b):
    dims, dtype, name, raw = [], 0, "", None
    for num, wt, v in _fields(b):
        if num == 1:
            dims.append(v) if wt == 0 else dims.extend(x for x in _unpack_varints(v))
        elif num == 2:
            dtype = v
        elif num == 8:
            name = bytes(v).decode()
        elif num == 9:
            raw = bytes(v)
    arr = np.frombuffer(raw, "<f4").reshape(dims).copy() if dtype == 1 and raw is not None else None
    return name, arr


def _unpack_varints(b):
    i = 0
    while i < len(b):
        v, i = _varint(b, i)
        yield v


def read_tensors(path):
    data = open(path, "rb").read()
    out = []          # (graph_id, name, array)
    counter = [0]

    def graph(b):
        gid = counter[0]; counter[0] += 1
        for num, wt, v in _fields(b):
            if num == 5 and wt == 2:
                name, arr = _tensor(v)
                if arr is not None:
                    out.append((gid, name, arr))
            elif num == 1 and wt == 2:
                outs, attrs = [], []
                for n2, w2, v2 in _fields(v):
                    if n2 == 2:
                        outs.append(bytes(v2).decode())
                    elif n2 == 5:
                        attrs.append(v2)
                for a in attrs:
                    for n3, w3, v3 in _fields(a):
                        if n3 == 5 and w3 == 2:
                            name, arr = _tensor(v3)
                            if arr is not None:
                                out.append((gid, name or (outs[0] if outs else ""), arr))
                        elif n3 == 6 and w3 == 2:
                            graph(v3)

    for num, wt, v in _fields(data):
        if num == 7 and wt == 2:
            graph(v)
    return out


def silero_16k_weights(path):
    """{short name: array} of the sub-graph that holds the [258, 1, 256] STFT basis (names stripped of the branch prefix)."""
    ts = read_tensors(path)
    g16 = [g for g, _, a in ts if a.shape == (258, 1, 256)][0]
    w = {}
    for g, name, a in ts:
        if g == g16:
            w[name.split("__Inline_0__")[-1]] = a
    return w
