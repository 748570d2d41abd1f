"""ctypes binding of libskw_engine.so (include/skw_engine.h) — the MI355X Whisper engine.

The library is the product; this module is plumbing for tests and bench.py.  It refuses to run
without the compiled HIP library: there is no Python / CPU fallback for any kernel.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKW_ENGINE_SO") or os.path.join(_HERE, "libskw_engine.so")      # (SKW_ENGINE_SO: A/B runs against another build of the same library, tools only)
_LIB = None


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_vocab", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_mels", "ftype")]


class FullParams(C.Structure):
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


class Timing(C.Structure):
    _fields_ = [("mel_ms", C.c_float), ("encode_ms", C.c_float), ("decode_ms", C.c_float), ("total_ms", C.c_float),
                ("n_windows", C.c_int32), ("n_decode_steps", C.c_int32), ("n_tokens", C.c_int32), ("n_row_steps", C.c_int32),
                ("decode_groups", C.c_int32), ("decode_group_rows", C.c_int32)]


class Trace(C.Structure):
    _fields_ = [("n", C.c_int32), ("steps", C.c_void_p)]


TRACE_DT = np.dtype([("chosen_id", "<i4"), ("forced_id", "<i4"), ("top1_id", "<i4"), ("top2_id", "<i4"),
                     ("top1", "<f4"), ("top2", "<f4"), ("forced_logit", "<f4"), ("lse", "<f4"), ("temperature", "<f4"), ("pad", "<i4")])      # skw_trace_step


class ResamplerState(C.Structure):
    _fields_ = [("last_index", C.c_double), ("ratio", C.c_double), ("chunk_frames", C.c_int32), ("channels", C.c_int32), ("hist", C.c_float * 32)]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libskw_engine.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.skw_device_count.restype = C.c_int
        L.skw_model_load.restype = C.c_void_p
        L.skw_model_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.skw_model_load_ex.restype = C.c_void_p
        L.skw_model_load_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.skw_model_quant_type.argtypes = [C.c_void_p]
        L.skw_model_free.argtypes = [C.c_void_p]
        L.skw_model_get_hparams.argtypes = [C.c_void_p, C.POINTER(HParams)]
        L.skw_model_token_text.restype = C.c_void_p
        L.skw_model_token_text.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.skw_model_lang_id.argtypes = [C.c_char_p]
        L.skw_ctx_create.restype = C.c_void_p
        L.skw_ctx_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.skw_ctx_free.argtypes = [C.c_void_p]
        L.skw_ctx_last_error.restype = C.c_char_p
        L.skw_ctx_last_error.argtypes = [C.c_void_p]
        L.skw_ctx_set_precision.argtypes = [C.c_void_p, C.c_int]
        L.skw_ctx_get_precision.argtypes = [C.c_void_p]
        L.skw_ctx_stream.restype = C.c_void_p
        L.skw_ctx_stream.argtypes = [C.c_void_p]
        L.skw_ctx_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.skw_full_default_params.argtypes = [C.POINTER(FullParams)]
        L.skw_full_batch.argtypes = [C.c_void_p, C.POINTER(FullParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(Result)]
        L.skw_full_batch_rng.argtypes = [C.c_void_p, C.POINTER(FullParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(Result)]
        L.skw_rng_state_init.argtypes = [C.c_void_p]
        L.skw_result_free.argtypes = [C.POINTER(Result)]
        L.skw_full_batch_traced.argtypes = [C.c_void_p, C.POINTER(FullParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_int,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(Trace), C.POINTER(Result)]
        L.skw_trace_free.argtypes = [C.POINTER(Trace)]
        L.skw_log_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.skw_conv_stem.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.skw_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.skw_decode_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.skw_dsp_create.restype = C.c_void_p
        L.skw_dsp_create.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.skw_dsp_free.argtypes = [C.c_void_p]
        L.skw_dsp_last_error.restype = C.c_char_p
        L.skw_dsp_last_error.argtypes = [C.c_void_p]
        L.skw_resampler_init.argtypes = [C.POINTER(ResamplerState), C.c_double, C.c_int, C.c_int]
        L.skw_resample_linear.argtypes = [C.c_void_p, C.POINTER(ResamplerState), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.skw_resample_polyphase.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
        L.skw_dsp_last_scan_fallback.argtypes = [C.c_void_p]
        L.skw_polyphase_stream_create.restype = C.c_void_p
        L.skw_polyphase_stream_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.skw_polyphase_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
        L.skw_polyphase_stream_free.argtypes = [C.c_void_p]
        L.skw_ctx_profile.argtypes = [C.c_void_p, C.c_int]
        L.skw_ctx_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.skw_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.skw_debug_sample_rows.argtypes = [C.c_void_p, C.POINTER(FullParams), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.skw_debug_enable.argtypes = [C.c_int]
        L.skw_debug_get.restype = C.c_long
        L.skw_debug_get.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        L.skw_ctx_kernel_clock.argtypes = [C.c_void_p, C.c_int]
        L.skw_ctx_kernel_clock_get.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.skw_ctx_kernel_clock_records.restype = C.c_long; L.skw_ctx_kernel_clock_records.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.skw_debug_xattn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.skw_debug_switch_name.restype = C.c_char_p; L.skw_debug_switch_name.argtypes = [C.c_int]
        L.skw_debug_switch_what.restype = C.c_char_p; L.skw_debug_switch_what.argtypes = [C.c_int]
        L.skw_debug_switch_default.argtypes = [C.c_int]
        L.skw_debug_switch_get.argtypes = [C.c_char_p]
        L.skw_debug_switch_set.argtypes = [C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


_TOKEN_DT = np.dtype([("id", "<i4"), ("tid", "<i4"), ("p", "<f4"), ("plog", "<f4"), ("pt", "<f4"), ("ptsum", "<f4"), ("margin", "<f4")])      # struct Token
_SEGMENT_DT = np.dtype([("t0", "<i8"), ("t1", "<i8"), ("tok_begin", "<i4"), ("tok_end", "<i4"), ("text_off", "<i4"), ("text_len", "<i4")])      # struct Segment
assert _TOKEN_DT.itemsize == C.sizeof(Token) and _SEGMENT_DT.itemsize == C.sizeof(Segment)


def _result_to_dict(r):
    text = C.string_at(r.text, r.text_len) if r.text else b""
    if r.n_tokens > 0:      # one structured view of the token array instead of five ctypes field reads per token (5 ms per 64 results in the bench's timed step)
        a = np.frombuffer((C.c_char * (r.n_tokens * C.sizeof(Token))).from_address(C.addressof(r.tokens.contents)), dtype=_TOKEN_DT)
        toks = list(zip(a["id"].tolist(), a["tid"].tolist(), a["p"].tolist(), a["plog"].tolist(), a["margin"].tolist()))
    else:
        toks = []
    segs = []
    if r.n_segments > 0:
        sg = np.frombuffer((C.c_char * (r.n_segments * C.sizeof(Segment))).from_address(C.addressof(r.segments.contents)), dtype=_SEGMENT_DT).tolist()
        segs = [dict(t0=t0, t1=t1, tokens=[t[0] for t in toks[b:e]], text=text[o:o + n]) for (t0, t1, b, e, o, n) in sg]
    return dict(segments=segs, tokens=toks, n_windows=r.n_windows, n_decode_steps=r.n_decode_steps,
                fallback_requested=r.fallback_requested, min_margin=r.min_margin, lang_id=r.lang_id)


def rng_state_new():
    """std::mt19937(0) as a uint32[625] (mt + index): the generator a freshly created whisper_state owns"""
    s = np.zeros(625, np.uint32)
    lib().skw_rng_state_init(s.ctypes.data)
    return s


class Model:
    def __init__(self, path, device=0, quant_mode=1):
        """quant_mode (block-quantised files): 1 = ggml's q8 arithmetic in the exact precision (SKW_QUANT_GGML), 0 = the dequantised f16 twin everywhere"""
        L = lib()
        err = C.create_string_buffer(512)
        self.h = L.skw_model_load_ex(path.encode(), device, int(quant_mode), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.quant = L.skw_model_quant_type(self.h)
        self.hp = HParams()
        L.skw_model_get_hparams(self.h, C.byref(self.hp))

    def token_bytes(self, i):
        n = C.c_int()
        p = lib().skw_model_token_text(self.h, i, C.byref(n))
        return C.string_at(p, n.value)

    def close(self):
        if self.h:
            lib().skw_model_free(self.h)
            self.h = None


class Context:
    def __init__(self, model, max_batch=1, max_samples=0):
        err = C.create_string_buffer(512)
        self.model = model
        self.h = lib().skw_ctx_create(model.h, max_batch, max_samples, err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("skw engine: " + lib().skw_ctx_last_error(self.h).decode())

    def last_error(self):
        return lib().skw_ctx_last_error(self.h).decode()

    PRECISIONS = {"exact": 0, "f16_mfma": 1}

    def set_precision(self, name):
        """'exact' (f32-chain contractions, bit-identical to the oracle) or 'f16_mfma' (f16 matrix cores: every greedy decision equals the exact
        mode's given the same history unless that step was a near-tie; a free-running transcript is identical up to its first near-tie)."""
        self._check(lib().skw_ctx_set_precision(self.h, self.PRECISIONS[name]))

    def get_precision(self):
        v = lib().skw_ctx_get_precision(self.h)
        return [k for k, x in self.PRECISIONS.items() if x == v][0]

    def default_params(self):
        p = FullParams()
        lib().skw_full_default_params(C.byref(p))
        return p

    def full_batch(self, clips, params=None, device_ptrs=None, n_samples=None, trace=False, forced=None, rng_states=None):
        """clips: list of 1-D float32 numpy arrays (host), or device_ptrs + n_samples for HBM-resident PCM.
        trace=True (or forced=[per-clip int32 id sequences]): skw_full_batch_traced — returns (results, traces), traces[i] a structured
        array (TRACE_DT) with one record per sampling decision of clip i; with `forced` the decoder is fed those ids (teacher forcing).
        rng_states=[uint32[625] or None per clip] (rng_state_new()): the temperature ladder's std::mt19937 stream of each clip's OWNER, continued and updated in place
        (skw_full_batch_rng: whisper.cpp keeps one generator per state and lets it run on across calls)."""
        p = params or self.default_params()
        if device_ptrs is not None:
            n = len(device_ptrs)
            ptrs = (C.c_void_p * n)(*device_ptrs)
            ns = (C.c_int32 * n)(*n_samples)
            on_dev = 1
        else:
            clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
            n = len(clips)
            ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in clips])
            ns = (C.c_int32 * n)(*[c.size for c in clips])
            on_dev = 0
        res = (Result * n)()
        if trace or forced is not None:
            tr = (Trace * n)()
            fptr = fn = None
            if forced is not None:
                keep = [np.ascontiguousarray(f, dtype=np.int32) for f in forced]
                fptr = (C.c_void_p * n)(*[k.ctypes.data for k in keep])
                fn = (C.c_int32 * n)(*[k.size for k in keep])
            self._check(lib().skw_full_batch_traced(self.h, C.byref(p), ptrs, ns, n, on_dev, fptr, fn, tr, res))
            traces = []
            for i in range(n):
                a = np.frombuffer((C.c_char * (tr[i].n * TRACE_DT.itemsize)).from_address(tr[i].steps), dtype=TRACE_DT).copy() if tr[i].n else np.zeros(0, TRACE_DT)
                traces.append(a)
                lib().skw_trace_free(C.byref(tr[i]))
        elif rng_states is not None:
            assert len(rng_states) == n and all(s is None or (s.dtype == np.uint32 and s.size == 625 and s.flags["C_CONTIGUOUS"]) for s in rng_states)
            sp = (C.c_void_p * n)(*[None if s is None else s.ctypes.data for s in rng_states])
            self._check(lib().skw_full_batch_rng(self.h, C.byref(p), ptrs, ns, n, on_dev, sp, res))
        else:
            self._check(lib().skw_full_batch(self.h, C.byref(p), ptrs, ns, n, on_dev, res))
        out = [_result_to_dict(res[i]) for i in range(n)]
        for i in range(n):
            lib().skw_result_free(C.byref(res[i]))
        if trace or forced is not None:
            return out, traces
        return out

    def timing(self):
        t = Timing()
        lib().skw_ctx_last_timing(self.h, C.byref(t))
        return {f: getattr(t, f) for f, _ in t._fields_}

    def profile(self, on=True):
        lib().skw_ctx_profile(self.h, 1 if on else 0)

    def profile_get(self):
        out = {}
        for cls in range(9):
            name = C.create_string_buffer(64); cnt = C.c_long(); ms = C.c_double(); fl = C.c_double(); by = C.c_double()
            if lib().skw_ctx_profile_get(self.h, cls, name, 64, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by)) == 0:
                out[name.value.decode()] = dict(count=cnt.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    def kernel_clock(self, on=True):
        """Arms / disarms the in-kernel launch clock of the decode step's cross attention (include/skw_engine.h, skw_ctx_kernel_clock)."""
        self._check(lib().skw_ctx_kernel_clock(self.h, 1 if on else 0))

    def kernel_clock_get(self):
        n = C.c_long(); su = C.c_double(); sl = C.c_double(); mn = C.c_double(); mx = C.c_double(); khz = C.c_int()
        self._check(lib().skw_ctx_kernel_clock_get(self.h, C.byref(n), C.byref(su), C.byref(sl), C.byref(mn), C.byref(mx), C.byref(khz)))
        return dict(launches=n.value, sum_us=su.value, sum_live_rows=sl.value, min_us=mn.value, max_us=mx.value, clock_khz=khz.value)

    def kernel_clock_records(self, cap=65536):
        """[(begin us, end us, live rows)] of every launch the last call recorded"""
        out = np.zeros((cap, 3), np.float64)
        n = lib().skw_ctx_kernel_clock_records(self.h, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("skw engine: " + self.last_error())
        return out[:n].copy()

    def stream(self):
        return lib().skw_ctx_stream(self.h)

    def log_mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        hp = self.model.hp
        cap = hp.n_mels * ((pcm.size + 480000) // 160 + 8)
        out = np.empty(cap, dtype=np.float32)
        n_len, n_org = C.c_int(), C.c_int()
        self._check(lib().skw_log_mel(self.h, pcm.ctypes.data, pcm.size, out.ctypes.data, cap, C.byref(n_len), C.byref(n_org)))
        return out[:hp.n_mels * n_len.value].reshape(hp.n_mels, n_len.value).copy(), n_org.value

    def conv_stem(self, pcm, seek=0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        hp = self.model.hp
        out = np.empty((hp.n_audio_ctx, hp.n_audio_state), dtype=np.float32)
        self._check(lib().skw_conv_stem(self.h, pcm.ctypes.data, pcm.size, seek, out.ctypes.data))
        return out

    def encode(self, pcm, seek=0, cross=True):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        hp = self.model.hp
        enc = np.empty((hp.n_audio_ctx, hp.n_audio_state), dtype=np.float32)
        ck = cv = None
        if cross:
            ck = np.empty((hp.n_text_layer, hp.n_audio_ctx, hp.n_text_state), dtype=np.float32)
            cv = np.empty_like(ck)
        self._check(lib().skw_encode(self.h, pcm.ctypes.data, pcm.size, seek, enc.ctypes.data,
                                     ck.ctypes.data if cross else None, cv.ctypes.data if cross else None))
        return enc, ck, cv

    def decode_logits(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.model.hp.n_vocab, dtype=np.float32)
        self._check(lib().skw_decode_logits(self.h, t.ctypes.data, t.size, out.ctypes.data))
        return out

    def sample_rows(self, hists, logits, params=None, form=0, want_filtered=False):
        """K11 on its own (skw_debug_sample_rows): one sampler launch decides the next token of len(hists) decoders whose sampled tokens so far are
        hists[r], on caller-supplied logits [rows][n_vocab].  form 0: the decode step's sampler; 1: the streaming kernel (leaves the filtered row
        in memory: want_filtered).  -> (tokens structured array, trace records, filtered logits or None)"""
        p = params or self.default_params()
        R = len(hists)
        stride = max(1, max(len(h) for h in hists))
        hb = np.zeros((R, stride), dtype=np.int32)
        for r, h in enumerate(hists):
            hb[r, :len(h)] = h
        nh = np.array([len(h) for h in hists], dtype=np.int32)
        lg = np.ascontiguousarray(logits, dtype=np.float32).reshape(R, self.model.hp.n_vocab)
        filt = np.empty_like(lg) if want_filtered else None
        toks = np.zeros(R, dtype=_TOKEN_DT)
        tr = np.zeros(R, dtype=TRACE_DT)
        self._check(lib().skw_debug_sample_rows(self.h, C.byref(p), R, hb.ctypes.data, stride, nh.ctypes.data, lg.ctypes.data, int(form),
                                                filt.ctypes.data if want_filtered else None, toks.ctypes.data, tr.ctypes.data))
        return toks, tr, filt

    def math(self, kind, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self._check(lib().skw_debug_math(self.h, kind, x.ctypes.data, out.ctypes.data, x.size))
        return out

    def close(self):
        if self.h:
            lib().skw_ctx_free(self.h)
            self.h = None


def switches():
    """The engine's switchboard (skw_kernels.h SkwSw / skw_engine.hip g_sw_defs): {name: (default, current, what)}.  No GPU needed."""
    L = lib()
    return {L.skw_debug_switch_name(i).decode(): (L.skw_debug_switch_default(i), L.skw_debug_switch_get(L.skw_debug_switch_name(i)), L.skw_debug_switch_what(i).decode())
            for i in range(L.skw_debug_switch_count())}


class switch:
    """`with engine.switch("GEMM16W", 0): ...` — flips one row of the switchboard in-process (what the environment variable SKW_GEMM16W=0 selects at start-up) and restores it."""
    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.old = lib().skw_debug_switch_get(self.name)
        if lib().skw_debug_switch_set(self.name, self.value) != 0:
            raise KeyError("no such switch: %s" % self.name.decode())
        return self

    def __exit__(self, *exc):
        lib().skw_debug_switch_set(self.name, self.old)
        return False


def debug_enable(on=True):
    lib().skw_debug_enable(1 if on else 0)


def debug_get(name):
    n = lib().skw_debug_get(name.encode(), None, 0)
    if n < 0:
        return None
    out = np.empty(n, dtype=np.float32)
    lib().skw_debug_get(name.encode(), out.ctypes.data, n)
    return out


class Dsp:
    """Model-free device context for the resampler kernels (include/skw_engine.h, R1-R3)."""

    def __init__(self, device=0):
        err = C.create_string_buffer(512)
        self.h = lib().skw_dsp_create(device, err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def linear_stream(self, ratio, chunk_frames, channels):
        st = ResamplerState()
        lib().skw_resampler_init(C.byref(st), ratio, chunk_frames, channels)
        return st

    def resample_linear(self, st, interleaved, n_chunks):
        x = np.ascontiguousarray(interleaved, dtype=np.float32)
        cap = int(n_chunks * st.chunk_frames * st.ratio) + 64
        out = np.empty(cap * st.channels, dtype=np.float32)
        n = C.c_int()
        if lib().skw_resample_linear(self.h, C.byref(st), x.ctypes.data, n_chunks, out.ctypes.data, cap, C.byref(n)) != 0:
            raise RuntimeError(lib().skw_dsp_last_error(self.h).decode())
        return out[:n.value * st.channels].copy()

    def last_scan_fallback(self):
        """1 when the last resample_linear needed the single-lane index walk (the parallel proposal failed its on-device check)."""
        return lib().skw_dsp_last_scan_fallback(self.h)

    def polyphase_stream(self, channels, in_rate, out_rate):
        return PolyphaseStream(self, channels, in_rate, out_rate)

    def resample_polyphase(self, interleaved, channels, in_rate, out_rate):
        x = np.ascontiguousarray(interleaved, dtype=np.float32)
        n_in = x.size // channels
        cap = n_in * out_rate // in_rate + 64
        out = np.empty(cap * channels, dtype=np.float32)
        n = C.c_long()
        if lib().skw_resample_polyphase(self.h, x.ctypes.data, n_in, channels, in_rate, out_rate, out.ctypes.data, cap, C.byref(n)) != 0:
            raise RuntimeError(lib().skw_dsp_last_error(self.h).decode())
        return out[:n.value * channels].copy()

    def close(self):
        if self.h:
            lib().skw_dsp_free(self.h)
            self.h = None


class PolyphaseStream:
    """Streaming polyphase resampler with its input tail resident on the device (skw_polyphase_stream_*)."""

    def __init__(self, dsp, channels, in_rate, out_rate):
        self.dsp, self.ch, self.in_rate, self.out_rate = dsp, channels, in_rate, out_rate
        self.h = lib().skw_polyphase_stream_create(dsp.h, channels, in_rate, out_rate)
        if not self.h:
            raise RuntimeError(lib().skw_dsp_last_error(dsp.h).decode())

    def push(self, interleaved, final=False):
        x = np.ascontiguousarray(interleaved if interleaved is not None else np.zeros(0, np.float32), dtype=np.float32)
        n_in = x.size // self.ch
        cap = (n_in + 4096) * self.out_rate // self.in_rate + 4096
        out = np.empty(cap * self.ch, dtype=np.float32)
        n = C.c_long()
        if lib().skw_polyphase_stream_push(self.h, x.ctypes.data if n_in else None, n_in, 1 if final else 0, out.ctypes.data, cap, C.byref(n)) != 0:
            raise RuntimeError(lib().skw_dsp_last_error(self.dsp.h).decode())
        return out[:n.value * self.ch].copy()

    def close(self):
        if self.h:
            lib().skw_polyphase_stream_free(self.h); self.h = None
