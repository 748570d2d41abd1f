"""Test-side access to the Silero gate: the product (libskw_vad.so, include/skw_vad.h) and the oracle (oracle/skw_silero_oracle.c,
fed by tests/onnx_mini.py), plus the seeded model file both read."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

import onnx_mini

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth_silero_path(seed=1234, lstm_op=False):
    path = "/tmp/skw_silero_%d%s.onnx" % (seed, "_lstmop" if lstm_op else "")
    if not os.path.exists(path):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "make_synth_silero.py"), path + ".tmp", "--seed", str(seed)] + (["--lstm-op"] if lstm_op else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        os.replace(path + ".tmp", path)
    return path


class _W(C.Structure):
    _fields_ = [("basis", C.c_void_p), ("cw", C.c_void_p * 4), ("cb", C.c_void_p * 4), ("w_ih", C.c_void_p), ("w_hh", C.c_void_p),
                ("b_ih", C.c_void_p), ("b_hh", C.c_void_p), ("ow", C.c_void_p), ("ob", C.c_float)]


class OracleSilero:
    """One stream through oracle/skw_silero_oracle.c with the call contract of vad.rs:67-120 (context and state kept here)."""

    def __init__(self, path):
        w = onnx_mini.silero_16k_weights(path)
        if "decoder.rnn.W" in w:       # ONNX LSTM operator blocks (i, o, f, c) -> LSTMCell blocks (i, f, g, o)
            def blk(a):
                return np.concatenate([a[128 * g:128 * (g + 1)] for g in (0, 2, 3, 1)])
            B = w["decoder.rnn.B"][0]
            w["decoder.rnn.weight_ih"], w["decoder.rnn.weight_hh"] = blk(w["decoder.rnn.W"][0]), blk(w["decoder.rnn.R"][0])
            w["decoder.rnn.bias_ih"], w["decoder.rnn.bias_hh"] = blk(B[:512]), blk(B[512:])
        self.w = {k: np.ascontiguousarray(v, np.float32) for k, v in w.items()}
        s = _W()
        s.basis = self.w["stft.forward_basis_buffer"].ctypes.data
        for l in range(4):
            s.cw[l] = self.w["encoder.%d.reparam_conv.weight" % l].ctypes.data
            s.cb[l] = self.w["encoder.%d.reparam_conv.bias" % l].ctypes.data
        s.w_ih, s.w_hh = self.w["decoder.rnn.weight_ih"].ctypes.data, self.w["decoder.rnn.weight_hh"].ctypes.data
        s.b_ih, s.b_hh = self.w["decoder.rnn.bias_ih"].ctypes.data, self.w["decoder.rnn.bias_hh"].ctypes.data
        s.ow = self.w["decoder.decoder.2.weight"].ctypes.data
        s.ob = float(self.w["decoder.decoder.2.bias"][0])
        self._s = s
        self._lib = C.CDLL(os.path.join(ROOT, "oracle", "libskw_oracle.so"))
        self._lib.skwo_silero_step.restype = C.c_float
        self._lib.skwo_silero_step.argtypes = [C.POINTER(_W), C.c_void_p, C.c_void_p]
        self.reset()

    def reset(self):
        self.state = np.zeros((2, 1, 128), np.float32)
        self.context = np.zeros(64, np.float32)

    def process_chunk(self, frame):
        frame = np.ascontiguousarray(frame, np.float32)
        assert frame.size == 512
        x = np.concatenate([self.context, frame])
        p = self._lib.skwo_silero_step(C.byref(self._s), x.ctypes.data, self.state.ctypes.data)
        self.context = frame[-64:].copy()
        return np.float32(p)


class ProductVad:
    def __init__(self, path):
        lib = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_vad.so"))
        lib.skw_vad_create.restype = C.c_void_p
        lib.skw_vad_create.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.skw_vad_process_chunk.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        lib.skw_vad_reset.argtypes = [C.c_void_p]
        lib.skw_vad_state.argtypes = [C.c_void_p, C.c_void_p]
        lib.skw_vad_free.argtypes = [C.c_void_p]
        self.lib = lib
        err = C.create_string_buffer(512)
        self.h = lib.skw_vad_create(path.encode(), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def process_chunk(self, frame):
        frame = np.ascontiguousarray(frame, np.float32)
        p = C.c_float()
        assert self.lib.skw_vad_process_chunk(self.h, frame.ctypes.data, C.byref(p)) == 0
        return np.float32(p.value)

    def state(self):
        s = np.zeros((2, 1, 128), np.float32)
        self.lib.skw_vad_state(self.h, s.ctypes.data)
        return s

    def reset(self):
        self.lib.skw_vad_reset(self.h)

    def close(self):
        if self.h:
            self.lib.skw_vad_free(self.h); self.h = None


def speechlike(n_frames, seed=0, pattern=((20, 0.0), (60, 0.2), (30, 0.0), (40, 0.05), (50, 0.0))):
    """n_frames x 512 samples: stretches of three-tone 'speech' at the given amplitude separated by near-silence (1e-4 noise)."""
    rng = np.random.default_rng(seed)
    out = np.zeros(n_frames * 512, np.float32)
    t = np.arange(n_frames * 512) / 16000.0
    tone = np.sin(2 * np.pi * 220 * t) + 0.6 * np.sin(2 * np.pi * 700 * t + 0.3) + 0.3 * np.sin(2 * np.pi * 2100 * t + 1.1)
    pos = 0
    while pos < n_frames:
        for n, amp in pattern:
            a, b = pos * 512, min(n_frames, pos + n) * 512
            out[a:b] = amp * tone[a:b]
            pos += n
            if pos >= n_frames:
                break
    out += (rng.random(out.size).astype(np.float32) - 0.5) * 2e-4
    return out.astype(np.float32)
