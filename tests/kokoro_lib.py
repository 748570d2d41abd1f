"""Test-side access to the Kokoro path: the product's synthesiser (libskw_tts.so, include/skw_tts.h) and the oracle
(oracle/skw_kokoro_oracle.c fed by tests/onnx_mini.py), a Python restatement of the text -> token ids step, and the seeded model
directory all of them read (tools/make_synth_kokoro.py)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

import onnx_mini

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U, BINS, STYLE, MAX_TOKENS, MAX_FRAMES = 120, 11, 128, 510, 6000


def synth_kokoro_dir(size="micro", seed=1234):
    path = "/tmp/skw_kokoro_%s_%d" % (size, seed)
    if not os.path.exists(os.path.join(path, "voices.bin")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_kokoro.py"), path + ".tmp", "--seed", str(seed), "--size", size])
        if os.path.exists(path):
            import shutil; shutil.rmtree(path)
        os.replace(path + ".tmp", path)
    return path


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


class _Dims(C.Structure):
    _fields_ = [("T", C.c_int), ("d", C.c_int), ("n_te", C.c_int), ("K", C.c_int), ("C", C.c_int), ("n_dec", C.c_int), ("G", C.c_int), ("scale", C.c_float), ("max_frames", C.c_int)]


class OracleTts:
    def __init__(self, model_dir):
        self.dir = model_dir
        self.w = {name: np.ascontiguousarray(a, np.float32) for _, name, a in onnx_mini.read_tensors(os.path.join(model_dir, "model.onnx"))}
        self.voices = np.fromfile(os.path.join(model_dir, "voices.bin"), "<f4").reshape(-1, MAX_TOKENS, 2 * STYLE)
        w = self.w
        self.d = w["text_encoder.embedding.weight"].shape[1]
        self.n_te = sum(1 for k in w if k.startswith("text_encoder.cnn.") and k.endswith(".weight"))
        self.n_dec = sum(1 for k in w if k.startswith("decoder.decode.") and k.endswith(".weight") and ".fc." not in k)
        self.K = w["predictor.duration_proj.weight"].shape[0]; self.Cc = w["decoder.encode.weight"].shape[0]; self.G = w["decoder.generator.ups.weight"].shape[1]
        order = ["text_encoder.embedding.weight"]
        for i in range(self.n_te):
            order += ["text_encoder.cnn.%d.%s" % (i, s) for s in ("weight", "bias", "norm.gamma", "norm.beta")]
        order += ["predictor.text_encoder.fc.weight", "predictor.text_encoder.fc.bias", "predictor.duration_proj.weight", "predictor.duration_proj.bias",
                  "predictor.F0_proj.weight", "predictor.F0_proj.style", "predictor.F0_proj.bias", "predictor.N_proj.weight", "predictor.N_proj.bias",
                  "decoder.encode.weight", "decoder.encode.bias", "decoder.encode.fc.weight", "decoder.encode.fc.bias"]
        for i in range(self.n_dec):
            order += ["decoder.decode.%d.%s" % (i, s) for s in ("weight", "bias", "fc.weight", "fc.bias")]
        order += ["decoder.generator.ups.weight", "decoder.generator.ups.bias", "decoder.generator.source.weight", "decoder.generator.resblock.alpha",
                  "decoder.generator.resblock.weight", "decoder.generator.resblock.bias", "decoder.generator.conv_post.weight", "decoder.generator.conv_post.bias"]
        self._ptrs = (C.c_void_p * len(order))(*[w[k].ctypes.data for k in order])
        self._lib = C.CDLL(os.path.join(ROOT, "oracle", "libskw_oracle.so"))
        self._lib.skwo_tts_synth.restype = C.c_long
        self._lib.skwo_tts_synth.argtypes = [C.POINTER(_Dims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)] + [C.c_void_p] * 5 + [C.c_long]

    def synth(self, text, sid=0, speed=1.0, length_scale=1.0):
        ids = np.asarray(tokenize(text, self.dir), np.int32); T = ids.size
        style = np.ascontiguousarray(self.voices[sid, max(0, min(T - 2, MAX_TOKENS - 1))])
        dims = _Dims(T, self.d, self.n_te, self.K, self.Cc, self.n_dec, self.G, np.float32(length_scale) / np.float32(speed), MAX_FRAMES)
        dur = np.zeros(T, np.int32); F = C.c_int()
        cap_f = MAX_FRAMES; cap_p = cap_f * U
        f0 = np.zeros(cap_f, np.float32); en = np.zeros(cap_f, np.float32); z = np.zeros(cap_f * self.Cc, np.float32); o = np.zeros(cap_p * 2 * BINS, np.float32); y = np.zeros(5 * cap_p, np.float32)
        n = self._lib.skwo_tts_synth(C.byref(dims), ids.ctypes.data, style.ctypes.data, self._ptrs, dur.ctypes.data, C.byref(F), f0.ctypes.data, en.ctypes.data, z.ctypes.data, o.ctypes.data, y.ctypes.data, y.size)
        assert n >= 0
        Fv = F.value
        return dict(ids=ids, dur=dur, f0=f0[:Fv].copy(), en=en[:Fv].copy(), z=z[:Fv * self.Cc].reshape(Fv, self.Cc).copy(), o=o[:Fv * U * 2 * BINS].reshape(Fv * U, 2 * BINS).copy(), y=y[:n].copy())


class _Cfg(C.Structure):
    _fields_ = [("model", C.c_char_p), ("voices", C.c_char_p), ("tokens", C.c_char_p), ("lexicon", C.c_char_p), ("length_scale", C.c_float), ("gpu_device", C.c_int32)]


class _Audio(C.Structure):
    _fields_ = [("samples", C.POINTER(C.c_float)), ("n", C.c_int32), ("sample_rate", C.c_int32)]


def tts_lib():
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_tts.so"))
    L.skw_tts_create.restype = C.c_void_p; L.skw_tts_create.argtypes = [C.POINTER(_Cfg), C.c_char_p, C.c_size_t]
    L.skw_tts_destroy.argtypes = [C.c_void_p]
    L.skw_tts_generate.restype = C.POINTER(_Audio); L.skw_tts_generate.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_float]
    L.skw_tts_destroy_audio.argtypes = [C.POINTER(_Audio)]
    L.skw_tts_last_error.restype = C.c_char_p; L.skw_tts_last_error.argtypes = [C.c_void_p]
    L.skw_tts_num_speakers.argtypes = [C.c_void_p]; L.skw_tts_sample_rate.argtypes = [C.c_void_p]
    L.skw_tts_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32]
    L.skw_tts_debug_get.restype = C.c_long; L.skw_tts_debug_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    L.skw_tts_last_ms.restype = C.c_float; L.skw_tts_last_ms.argtypes = [C.c_void_p]
    return L


class Tts:
    """The product: libskw_tts.so through its C ABI (the calls a Rust host binds in place of sherpa-onnx's)."""

    def __init__(self, model_dir, device=0):
        self.L = tts_lib()
        d = model_dir
        cfg = _Cfg((d + "/model.onnx").encode(), (d + "/voices.bin").encode(), (d + "/tokens.txt").encode(), (d + "/lexicon-us-en.txt," + d + "/lexicon-zh.txt").encode(), 1.0, device)
        err = C.create_string_buffer(512)
        self.h = self.L.skw_tts_create(C.byref(cfg), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def tokenize(self, text):
        ids = np.zeros(MAX_TOKENS, np.int32)
        n = self.L.skw_tts_tokenize(self.h, text.encode(), ids.ctypes.data, ids.size)
        return ids[:n].copy()

    def generate(self, text, sid=0, speed=1.0):
        a = self.L.skw_tts_generate(self.h, text.encode(), sid, speed)
        if not a:
            raise RuntimeError(self.L.skw_tts_last_error(self.h).decode())
        y = np.ctypeslib.as_array(a.contents.samples, shape=(a.contents.n,)).copy(); rate = a.contents.sample_rate
        self.L.skw_tts_destroy_audio(a)
        return y, rate

    def tap(self, what):
        n = self.L.skw_tts_debug_get(self.h, what, None, 0)
        out = np.zeros(n, np.float32); self.L.skw_tts_debug_get(self.h, what, out.ctypes.data, n)
        return out

    def last_ms(self):
        return self.L.skw_tts_last_ms(self.h)

    def close(self):
        if self.h:
            self.L.skw_tts_destroy(self.h); self.h = None


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
