"""Test-side access to the Kokoro path: the product's synthesiser (streamkit_amd/tts.py: libskw_tts.so, include/skw_tts.h) and the oracle
(oracle/skw_kokoro_oracle.cpp fed by tests/onnx_mini.py), a Python restatement of the text -> token ids step, and the seeded model
directory all of them read (tools/make_synth_kokoro.py)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

import onnx_mini

from streamkit_amd.tts import BINS, MAX_FRAMES, MAX_TOKENS, STYLE, Tts, synth_kokoro_dir, tts_lib  # noqa: F401  (the product's binding; re-exported for the tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tokens(model_dir):
    m = {}
    for line in open(os.path.join(model_dir, "tokens.txt"), encoding="utf-8").read().split("\n"):
        if not line:
            continue
        sym, _, idx = line.rpartition(" ")
        m[sym or " "] = int(idx)
    return m


def load_lexicon(model_dir, sym2id):
    lex = {}
    p = os.path.join(model_dir, "lexicon-us-en.txt")
    if os.path.exists(p):
        for line in open(p, encoding="utf-8").read().split("\n"):
            parts = line.split()
            if len(parts) >= 2 and parts[0].lower() not in lex:
                ids = [sym2id[ch] for ph in parts[1:] for ch in ph if ch in sym2id]
                if ids:
                    lex[parts[0].lower()] = ids
    return lex


def tokenize(text, model_dir):
    """text -> ids (include/skw_tts.h: lexicon words -> their phoneme ids, else code point by code point through tokens.txt; pad 0 both ends)."""
    sym2id = load_tokens(model_dir); lex = load_lexicon(model_dir, sym2id)
    ids = [0]; i = 0
    while i < len(text) and len(ids) < MAX_TOKENS - 1:
        j = i
        while j < len(text) and (text[j].isascii() and (text[j].isalpha() or text[j] == "'")):
            j += 1
        word = text[i:j]
        if word and word.lower() in lex:
            for t in lex[word.lower()]:
                if len(ids) < MAX_TOKENS - 1:
                    ids.append(t)
            i = j; continue
        ch = text[i]; i += 1
        if ch in sym2id:
            ids.append(sym2id[ch])
        elif "A" <= ch <= "Z" and ch.lower() in sym2id:
            ids.append(sym2id[ch.lower()])
    ids.append(0)
    return ids


def style_row(T):
    """voices.bin row for T token ids (pad id at both ends): the published pipeline's pack[len(phonemes) - 1] = T - 3"""
    return max(0, min(T - 3, MAX_TOKENS - 1))


def unit_noise(counter):
    """numpy restatement of the product's stand-in for torch.randn in the harmonic source (include/skw_kokoro_net.h unit_noise: a splitmix64 counter hash, top 24 bits, uniform in
    [-sqrt 3, sqrt 3)); sample n, harmonic h (1-based) draws unit_noise(16 n + h)"""
    with np.errstate(over="ignore"):
        x = np.asarray(counter).astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float32)
    return ((u * np.float32(1.0 / 8388608.0) - np.float32(1.0)) * np.float32(1.7320508075688772)).astype(np.float32)


def source_noise(L, harmonics=9):
    """[1, L, harmonics] torch tensor: what the published SineGen draws with randn_like, as the product draws it"""
    import torch
    n = np.arange(L, dtype=np.uint64)[:, None] * np.uint64(16) + np.arange(1, harmonics + 1, dtype=np.uint64)[None, :]
    return torch.from_numpy(unit_noise(n)).unsqueeze(0)


def load_model_tensors(model_dir):
    return {name: np.ascontiguousarray(a, np.float32) for _, name, a in onnx_mini.read_tensors(os.path.join(model_dir, "model.onnx"))}


def load_voices(model_dir):
    return np.fromfile(os.path.join(model_dir, "voices.bin"), "<f4").reshape(-1, MAX_TOKENS, 2 * STYLE)


class _TensorRef(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("n_dims", C.c_int32), ("dims", C.c_int64 * 4)]


class OracleTts:
    """oracle/skw_kokoro_oracle.cpp: the Kokoro network (include/skw_kokoro_net.h) with every operator a CPU loop, fed the tensors tests/onnx_mini.py reads from model.onnx"""

    def __init__(self, model_dir):
        self.dir = model_dir
        self.w = {name: np.ascontiguousarray(a, np.float32) for _, name, a in onnx_mini.read_tensors(os.path.join(model_dir, "model.onnx"))}
        self.voices = np.fromfile(os.path.join(model_dir, "voices.bin"), "<f4").reshape(-1, MAX_TOKENS, 2 * STYLE)
        self.hid = self.w["bert_encoder.weight"].shape[1]; self.d = self.w["bert_encoder.weight"].shape[0]
        self.gen_c0 = self.w["decoder.generator.ups.0.weight"].shape[0]
        names = sorted(self.w)
        self._keep = [n.encode() for n in names]
        self._refs = (_TensorRef * len(names))()
        for i, n in enumerate(names):
            a = self.w[n]; r = self._refs[i]
            r.name = self._keep[i]; r.data = a.ctypes.data; r.n_dims = a.ndim
            for k in range(a.ndim):
                r.dims[k] = a.shape[k]
        self._lib = C.CDLL(os.path.join(ROOT, "oracle", "libskw_oracle.so"))
        self._lib.skwo_kokoro_forward.restype = C.c_long
        self._lib.skwo_kokoro_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int32)] + [C.c_void_p] * 6 + \
                                                [C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_char_p, C.c_int]

    def synth(self, text, sid=0, speed=1.0, length_scale=1.0, ids=None):
        ids = np.asarray(tokenize(text, self.dir) if ids is None else ids, np.int32); T = ids.size
        style = np.ascontiguousarray(self.voices[sid, style_row(T)])
        dur = np.zeros(T, np.int32); F = C.c_int32()
        cap_f = MAX_FRAMES
        bert = np.zeros(T * self.hid, np.float32); d_en = np.zeros(T * self.d, np.float32); t_en = np.zeros(T * self.d, np.float32)
        f0 = np.zeros(2 * cap_f, np.float32); en = np.zeros(2 * cap_f, np.float32)
        dec = np.zeros(2 * cap_f * self.gen_c0, np.float32); post = np.zeros((2 * cap_f * 60 + 1) * 22, np.float32); y = np.zeros(600 * cap_f, np.float32); har = np.zeros_like(post)
        err = C.create_string_buffer(512)
        n = self._lib.skwo_kokoro_forward(self._refs, len(self._refs), ids.ctypes.data, T, style.ctypes.data, np.float32(length_scale) / np.float32(speed), cap_f,
                                          dur.ctypes.data, C.byref(F), bert.ctypes.data, d_en.ctypes.data, t_en.ctypes.data, f0.ctypes.data, en.ctypes.data,
                                          dec.ctypes.data, dec.size, post.ctypes.data, post.size, y.ctypes.data, y.size, har.ctypes.data, har.size, err, 512)
        if n < 0:
            raise RuntimeError(err.value.decode())
        Fv = F.value
        return dict(ids=ids, dur=dur, F=Fv, bert=bert.reshape(T, self.hid), d_en=d_en.reshape(T, self.d), t_en=t_en.reshape(T, self.d), f0=f0[:2 * Fv].copy(), en=en[:2 * Fv].copy(),
                    dec=dec[:2 * Fv * self.gen_c0].reshape(2 * Fv, self.gen_c0).copy(), post=post[:(2 * Fv * 60 + 1) * 22].reshape(-1, 22).copy(), har=har[:(2 * Fv * 60 + 1) * 22].reshape(-1, 22).copy(), y=y[:n].copy())


# ---- Python restatement of the node's text front end (kokoro_node.rs:444-492, 696-731; sentence_splitter.rs:15-58), independent of skw_kokoro_text.h
_WS = set([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]) | set(range(0x2000, 0x200B))
_KEEP = set(ord(c) for c in " .,!?-'\"\n:;。，！？、；：（）")


def _keeps(c):
    return (0x61 <= c <= 0x7A) or (0x41 <= c <= 0x5A) or (0x30 <= c <= 0x39) or c in _KEEP or (0xE0 <= c <= 0xFF) or (0xC0 <= c <= 0x178) or (0x4E00 <= c <= 0x9FFF)


def sanitize_text(text):
    kept = "".join(ch if _keeps(ord(ch)) else (" " if ord(ch) in _WS else "") for ch in text)
    words, cur = [], ""
    for ch in kept:
        if ord(ch) in _WS:
            if cur:
                words.append(cur); cur = ""
        else:
            cur += ch
    if cur:
        words.append(cur)
    return " ".join(words)


_FINAL = (".", "!", "?", "。", "！", "？")
_BOUNDS = [". ", ".\n", "! ", "!\n", "? ", "?\n", "。", "！", "？"]


def _trim(s):
    a, b = 0, len(s)
    while a < b and ord(s[a]) in _WS:
        a += 1
    while b > a and ord(s[b - 1]) in _WS:
        b -= 1
    return s[a:b]


def extract_sentence(buf, min_length):
    """-> (sentence or None, remaining buffer); lengths in UTF-8 bytes"""
    if len(buf.encode()) < min_length:
        return None, buf
    for b in _BOUNDS:
        pos = buf.find(b)
        if pos >= 0:
            end = pos + len(b)
            return _trim(buf[:end]), buf[end:]
    if buf.endswith(_FINAL):
        return buf, ""
    return None, buf


def node_sentences(texts, min_length=10, flush=True):
    """The sentences a Kokoro node speaks for a sequence of input texts (process per packet, then flush)."""
    out, buf = [], ""
    for t in texts:
        s = sanitize_text(t)
        if not s:
            continue
        if not s.endswith(_FINAL):
            s += "."
        buf += s
        while True:
            sent, buf = extract_sentence(buf, min_length)
            if sent is None:
                break
            out.append(sent)
    if flush and buf:
        out.append(buf)
    return out
