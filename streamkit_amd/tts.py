"""ctypes binding of libskw_tts.so (include/skw_tts.h: the calls a Rust host binds in place of sherpa-onnx's, kokoro/src/ffi.rs:119-137) and the seeded Kokoro model
directory tools/make_synth_kokoro.py writes.  Product-side access only: the CPU checkers live under tests/ and oracle/."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINS, STYLE, MAX_TOKENS, MAX_FRAMES = 11, 128, 510, 3000


def synth_kokoro_dir(size="micro", seed=1234):
    path = "/tmp/skw_kokoro_%s_%d" % (size, seed)
    if not os.path.exists(os.path.join(path, "voices.bin")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synth_kokoro.py"), path + ".tmp", "--seed", str(seed), "--size", size])
        if os.path.exists(path):
            import shutil; shutil.rmtree(path)
        os.replace(path + ".tmp", path)
    return path


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
    L.skw_tts_generate_ids.restype = C.POINTER(_Audio); L.skw_tts_generate_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float]
    L.skw_tts_debug_enable.argtypes = [C.c_void_p, C.c_int]; L.skw_tts_debug_conv_mode.argtypes = [C.c_int]; L.skw_tts_debug_lstm_mode.argtypes = [C.c_int]
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

    def generate(self, text, sid=0, speed=1.0, ids=None):
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int32)
            a = self.L.skw_tts_generate_ids(self.h, ids.ctypes.data, ids.size, sid, speed)
        else:
            a = self.L.skw_tts_generate(self.h, text.encode(), sid, speed)
        if not a:
            raise RuntimeError(self.L.skw_tts_last_error(self.h).decode())
        y = np.ctypeslib.as_array(a.contents.samples, shape=(a.contents.n,)).copy(); rate = a.contents.sample_rate
        self.L.skw_tts_destroy_audio(a)
        return y, rate

    def taps(self, on=True):
        self.L.skw_tts_debug_enable(self.h, 1 if on else 0)

    def tap(self, what):
        n = self.L.skw_tts_debug_get(self.h, what, None, 0)
        out = np.zeros(n, np.float32); self.L.skw_tts_debug_get(self.h, what, out.ctypes.data, n)
        return out

    def last_ms(self):
        return self.L.skw_tts_last_ms(self.h)

    def close(self):
        if self.h:
            self.L.skw_tts_destroy(self.h); self.h = None
