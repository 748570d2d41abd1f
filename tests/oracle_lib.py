"""ctypes binding of oracle/libskw_oracle.so — test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_vocab", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_mels", "ftype")]


class Params(C.Structure):
    _fields_ = [("lang_id", C.c_int32), ("translate", C.c_int32), ("suppress_blank", C.c_int32), ("suppress_nst", C.c_int32),
                ("no_timestamps", C.c_int32), ("single_segment", C.c_int32), ("max_tokens", C.c_int32),
                ("max_initial_ts", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
                ("no_speech_thold", C.c_float), ("n_threads", C.c_int32),
                ("temperature", C.c_float), ("temperature_inc", C.c_float)]


class Segment(C.Structure):
    _fields_ = [("t0", C.c_int64), ("t1", C.c_int64), ("tok_begin", C.c_int32), ("tok_end", C.c_int32),
                ("text_off", C.c_int32), ("text_len", C.c_int32)]


class Token(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float), ("margin", C.c_float)]


class Result(C.Structure):
    _fields_ = [("n_segments", C.c_int32), ("n_tokens", C.c_int32), ("n_windows", C.c_int32), ("n_decode_steps", C.c_int32),
                ("fallback_requested", C.c_int32), ("min_margin", C.c_float),
                ("segments", C.POINTER(Segment)), ("tokens", C.POINTER(Token)), ("text", C.c_void_p), ("text_len", C.c_int32), ("lang_id", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "libskw_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.skwo_load.restype = C.c_void_p
        L.skwo_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.skwo_free.argtypes = [C.c_void_p]
        L.skwo_set_quant_mode.argtypes = [C.c_int]
        L.skwo_model_quant.argtypes = [C.c_void_p]
        L.skwo_debug_linear_q8.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.skwo_get_hparams.argtypes = [C.c_void_p, C.POINTER(HParams)]
        L.skwo_token_str.restype = C.c_void_p
        L.skwo_token_str.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.skwo_default_params.argtypes = [C.POINTER(Params)]
        L.skwo_log_mel.restype = C.POINTER(C.c_float)
        L.skwo_log_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.skwo_free_buf.argtypes = [C.c_void_p]
        L.skwo_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.skwo_conv_stem.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.skwo_dec_new.restype = C.c_void_p
        L.skwo_dec_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.skwo_dec_free.argtypes = [C.c_void_p]
        L.skwo_dec_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.skwo_full.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int, C.POINTER(Result)]
        L.skwo_full_rng.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int, C.POINTER(Result), C.c_void_p]
        L.skwo_result_free.argtypes = [C.POINTER(Result)]
        L.skwo_resampler_new.restype = C.c_void_p
        L.skwo_resampler_new.argtypes = [C.c_double, C.c_int, C.c_int]
        L.skwo_resampler_free.argtypes = [C.c_void_p]
        L.skwo_resampler_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.skwo_segment_sim.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_float, C.c_void_p, C.c_int]
        L.skwo_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.skwo_debug_enable.argtypes = [C.c_int]
        L.skwo_mfma_f16_element.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.skwo_mfma_f16_element.restype = C.c_float
        L.skwo_mfma_f16_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.skwo_mfma_f16_tiles.restype = None
        L.skwo_mfma_f16_elements.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.skwo_mfma_f16_elements.restype = None
        L.skwo_gemm_f16mfma.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.skwo_debug_get.restype = C.c_long
        L.skwo_debug_get.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        _LIB = L
    return _LIB


class OracleModel:
    def __init__(self, path, quant_mode=1):
        """quant_mode: 1 = ggml's q8 arithmetic for a uniformly quantised file (the exact precision of the engine), 0 = its dequantised f16 twin"""
        L = lib()
        err = C.create_string_buffer(256)
        L.skwo_set_quant_mode(int(quant_mode))
        self.h = L.skwo_load(path.encode(), err, 256)
        L.skwo_set_quant_mode(1)
        self.quant = L.skwo_model_quant(self.h) if self.h else 0
        if not self.h:
            raise RuntimeError("oracle load failed: " + err.value.decode())
        self.hp = HParams()
        L.skwo_get_hparams(self.h, C.byref(self.hp))

    def close(self):
        if self.h:
            lib().skwo_free(self.h)
            self.h = None

    def token_bytes(self, i):
        n = C.c_int()
        p = lib().skwo_token_str(self.h, i, C.byref(n))
        return C.string_at(p, n.value)

    def math(self, kind, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        lib().skwo_math(self.h, kind, x.ctypes.data, out.ctypes.data, x.size)
        return out

    def default_params(self):
        p = Params()
        lib().skwo_default_params(C.byref(p))
        return p

    def log_mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n_len, n_org = C.c_int(), C.c_int()
        ptr = lib().skwo_log_mel(self.h, pcm.ctypes.data, pcm.size, C.byref(n_len), C.byref(n_org))
        mel = np.ctypeslib.as_array(ptr, shape=(self.hp.n_mels, n_len.value)).copy()
        lib().skwo_free_buf(ptr)
        return mel, n_org.value

    def conv_stem(self, mel, seek=0, n_threads=0):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        out = np.empty((self.hp.n_audio_ctx, self.hp.n_audio_state), dtype=np.float32)
        lib().skwo_conv_stem(self.h, mel.ctypes.data, mel.shape[1], seek, n_threads, out.ctypes.data)
        return out

    def encode(self, mel, seek=0, n_threads=0, cross=True):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        hp = self.hp
        enc = np.empty((hp.n_audio_ctx, hp.n_audio_state), dtype=np.float32)
        ck = cv = None
        if cross:
            ck = np.empty((hp.n_text_layer, hp.n_audio_ctx, hp.n_text_state), dtype=np.float32)
            cv = np.empty_like(ck)
        lib().skwo_encode(self.h, mel.ctypes.data, mel.shape[1], seek, n_threads, enc.ctypes.data,
                          ck.ctypes.data if cross else None, cv.ctypes.data if cross else None)
        return enc, ck, cv

    def decoder(self, ck, cv):
        return OracleDecoder(self, ck, cv)

    def full(self, pcm, params=None, rng_state=None):
        """rng_state: a uint32[625] std::mt19937 state (mt + index) that the call continues and updates in place — whisper.cpp's per-state generator; None: seeded with 0"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        p = params or self.default_params()
        r = Result()
        if rng_state is not None:
            assert rng_state.dtype == np.uint32 and rng_state.size == 625 and rng_state.flags["C_CONTIGUOUS"]
            rc = lib().skwo_full_rng(self.h, C.byref(p), pcm.ctypes.data, pcm.size, C.byref(r), rng_state.ctypes.data)
        else:
            rc = lib().skwo_full(self.h, C.byref(p), pcm.ctypes.data, pcm.size, C.byref(r))
        if rc != 0:
            raise RuntimeError("skwo_full rc=%d" % rc)
        text = C.string_at(r.text, r.text_len) if r.text else b""
        toks = [(r.tokens[i].id, r.tokens[i].tid, r.tokens[i].p, r.tokens[i].plog, r.tokens[i].margin) for i in range(r.n_tokens)]
        segs = [dict(t0=r.segments[i].t0, t1=r.segments[i].t1, tokens=[t[0] for t in toks[r.segments[i].tok_begin:r.segments[i].tok_end]],
                     text=text[r.segments[i].text_off:r.segments[i].text_off + r.segments[i].text_len]) for i in range(r.n_segments)]
        out = dict(segments=segs, tokens=toks, n_windows=r.n_windows, n_decode_steps=r.n_decode_steps,
                   fallback_requested=r.fallback_requested, min_margin=r.min_margin, lang_id=r.lang_id)
        lib().skwo_result_free(C.byref(r))
        return out


class OracleDecoder:
    def __init__(self, model, ck, cv):
        self.model = model
        self.ck = np.ascontiguousarray(ck, dtype=np.float32)
        self.cv = np.ascontiguousarray(cv, dtype=np.float32)
        self.h = lib().skwo_dec_new(model.h, self.ck.ctypes.data, self.cv.ctypes.data)

    def step(self, tokens, n_past, n_threads=0):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        logits = np.empty(self.model.hp.n_vocab, dtype=np.float32)
        rc = lib().skwo_dec_step(self.h, t.ctypes.data, t.size, n_past, n_threads, logits.ctypes.data)
        if rc != 0:
            raise RuntimeError("dec_step rc=%d" % rc)
        return logits

    def close(self):
        if self.h:
            lib().skwo_dec_free(self.h)
            self.h = None


def debug_enable(on=True):
    lib().skwo_debug_enable(1 if on else 0)


def debug_get(name):
    n = lib().skwo_debug_get(name.encode(), None, 0)
    if n < 0:
        return None
    out = np.empty(n, dtype=np.float32)
    lib().skwo_debug_get(name.encode(), out.ctypes.data, n)
    return out


def segment_sim(prob, threshold=0.5, min_silence_ms=700, max_secs=30.0, max_cuts=256):
    prob = np.ascontiguousarray(prob, dtype=np.float32)
    cuts = np.zeros((max_cuts, 6), dtype=np.int64)
    n = lib().skwo_segment_sim(prob.ctypes.data, prob.size, threshold, min_silence_ms, max_secs, cuts.ctypes.data, max_cuts)
    return cuts[:min(n, max_cuts)].tolist()


class OracleResampler:
    """rubato FastFixedIn<f32>/Linear restatement (planar in, planar out)."""

    def __init__(self, ratio, chunk_frames, channels):
        self.h = lib().skwo_resampler_new(ratio, chunk_frames, channels)
        self.chunk, self.ch, self.ratio = chunk_frames, channels, ratio

    def process(self, planar):
        planar = np.ascontiguousarray(planar, dtype=np.float32).reshape(self.ch, self.chunk)
        cap = int(self.chunk * max(4.0, 1.1 * self.ratio)) + 64
        out = np.zeros((self.ch, cap), dtype=np.float32)
        n = lib().skwo_resampler_process(self.h, planar.ctypes.data, out.ctypes.data, cap)
        assert n >= 0
        return out[:, :n].copy()


def mfma_f16_tiles(A, B, Cc):
    """P instructions of v_mfma_f32_16x16x32_f16 as oracle/ restates them: A [P][16][32] u16, B [P][32][16] u16, C [P][16][16] f32 -> D."""
    A = np.ascontiguousarray(A, np.uint16); B = np.ascontiguousarray(B, np.uint16); Cc = np.ascontiguousarray(Cc, np.float32)
    P = A.shape[0]
    assert A.shape == (P, 16, 32) and B.shape == (P, 32, 16) and Cc.shape == (P, 16, 16)
    D = np.empty((P, 16, 16), np.float32)
    lib().skwo_mfma_f16_tiles(A.ctypes.data, B.ctypes.data, Cc.ctypes.data, D.ctypes.data, P)
    return D


def mfma_f16_elements(a, b, c):
    """n single output elements: a, b [n][32] u16 in slot order, c [n] f32 -> d [n]."""
    a = np.ascontiguousarray(a, np.uint16); b = np.ascontiguousarray(b, np.uint16); c = np.ascontiguousarray(c, np.float32)
    assert a.shape == b.shape == (len(c), 32)
    d = np.empty(len(c), np.float32)
    lib().skwo_mfma_f16_elements(a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, len(c))
    return d


def gemm_f16mfma(A, W, n_split=1):
    """C[m][n] = the matrix cores' contraction of A[m][:] (u16 f16 bits, [M][K]) and W[n][:] ([N][K]) in the f16_mfma GEMM kernels' order (skwo_gemm_f16mfma)."""
    A = np.ascontiguousarray(A, np.uint16); W = np.ascontiguousarray(W, np.uint16)
    M, K = A.shape; N = W.shape[0]
    assert W.shape[1] == K
    out = np.empty((M, N), np.float32)
    r = lib().skwo_gemm_f16mfma(A.ctypes.data, K, W.ctypes.data, K, M, N, K, n_split, out.ctypes.data, N)
    if r != 0:
        raise ValueError("skwo_gemm_f16mfma refused M %d N %d K %d n_split %d" % (M, N, K, n_split))
    return out
