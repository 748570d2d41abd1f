"""K11 (whisper_process_logits) on its own: seeded cases, the oracle's side, and an INDEPENDENT implementation of the same rules —
transformers' Whisper logits processors (WhisperTimeStampLogitsProcessor, SuppressTokensLogitsProcessor,
SuppressTokensAtBeginLogitsProcessor: HF's restatement of openai/whisper decoding.py, which is also what whisper.cpp's comments cite).
Test infrastructure only; transformers is a tool of the build container and is not needed on the GPU box (the committed fixture
tests/golden/logit_rule_cases.json carries the checked outcomes there).

Rules whisper.cpp has and HF lacks / states differently (each handled explicitly, never papered over):
  D-a  first decision of a window: openai/whisper and HF suppress every non-timestamp token ("suppress generating non-timestamp tokens at
       the beginning"); whisper.cpp has no such line — its first token is a timestamp only through the timestamp-mass rule and max_initial_ts.
  D-b  "timestamps shouldn't decrease": HF forbids [timestamp_begin, last) after a closed pair's first member and [timestamp_begin, last]
       otherwise; whisper.cpp forbids [token_beg, token_beg + seek_delta/2) whenever has_ts — the last timestamp itself stays admissible, and
       a lone <|0.00|> (id == token_beg) never sets has_ts.  So HF additionally suppresses {last timestamp} when the last token is text.
  D-c  suppress_blank / the static specials / suppress_nst are id lists on both sides (HF: SuppressTokens*): the lists are whisper.cpp's
       (string look-ups in the model's vocabulary), handed to HF as ids.
  D-d  temperature: whisper.cpp divides the logits before the filters; HF's TemperatureLogitsWarper runs after the processors.  Division by a
       positive constant commutes with every rule except through rounding of the mass rule's comparison; cases here run at temperature 0.
  D-f  when the timestamp-mass rule fires, whisper.cpp sets the text entries of logits and logprobs to -inf WITHOUT renormalising: plog / p of the
       chosen timestamp stay relative to the distribution before the text was removed (HF's downstream log_softmax renormalises).  Argmax unaffected.
  D-e  no_speech_prob: softmax of the unfiltered first-step logits at the nospeech token on both sides.
With D-a / D-b's extra tokens removed from the INPUT of both sides (raw logit -inf: suppressed by construction everywhere), the two
implementations must produce the same admissible set, the same timestamp-mass decision and the same argmax on every case."""
import ctypes as C
import hashlib

import numpy as np

from oracle_lib import lib, Params, Token

N_PROMPT = 3      # sot, language, task: HF's begin_index for a multilingual model with timestamps


def bind():
    L = lib()
    L.skwo_debug_process_logits.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int,
                                            C.c_void_p, C.c_void_p, C.POINTER(Token), C.POINTER(C.c_float)]
    L.skwo_debug_rule_ids.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return L


def rule_ids(om, kind):
    L = bind()
    n = L.skwo_debug_rule_ids(om.h, kind, None, 0)
    a = np.zeros(n, dtype=np.int32)
    L.skwo_debug_rule_ids(om.h, kind, a.ctypes.data, n)
    return [int(x) for x in a]


def special_ids(om):
    eot, tnot, beg, nosp, sot = rule_ids(om, 3)
    return dict(eot=eot, not_=tnot, beg=beg, nosp=nosp, sot=sot)


def oracle_process(om, params, hist, raw, flags=0):
    """-> (filtered logits, logprobs, chosen Token fields, no_speech_prob) or None when the token loop could not have produced hist"""
    L = bind()
    NV = om.hp.n_vocab
    h = np.ascontiguousarray(hist, dtype=np.int32)
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    out = np.empty(NV, dtype=np.float32)
    lp = np.empty(NV, dtype=np.float32)
    tk = Token()
    nsp = C.c_float()
    rc = L.skwo_debug_process_logits(om.h, C.byref(params), h.ctypes.data, h.size, raw.ctypes.data, 0.0, flags, out.ctypes.data, lp.ctypes.data, C.byref(tk), C.byref(nsp))
    if rc != 0:
        return None
    return out, lp, dict(id=tk.id, tid=tk.tid, p=tk.p, plog=tk.plog, pt=tk.pt, ptsum=tk.ptsum, margin=tk.margin), nsp.value


def make_case(rng, sp, NV, kind):
    """A decoder history the token loop can produce + raw logits that exercise the rule the kind names.  Returns (hist, raw)."""
    beg, eot = sp["beg"], sp["eot"]
    n_ts = NV - beg
    hist = []
    if kind != "initial":
        # <|t0|> text.. <|t1|><|t1'|> text.. ... ; then cut at a random point so that every grammar position is visited
        t = int(rng.integers(0, 40))
        hist.append(beg + t)
        n_seg = int(rng.integers(1, 4))
        for _ in range(n_seg):
            hist += [int(x) for x in rng.integers(0, eot, size=int(rng.integers(1, 9)))]
            t = min(t + int(rng.integers(0, 300)), n_ts - 1)
            hist.append(beg + t)
            t2 = min(t + int(rng.integers(0, 3)), n_ts - 1)       # the pair's second member: usually the same timestamp
            hist.append(beg + t2)
            t = t2
        k_any = int(rng.integers(1, len(hist) + 1))
        if kind == "after_pair":
            k = len(hist)
        elif kind == "after_lone_ts":
            k = len(hist) - 1
        elif kind == "text":
            idx = [i for i, x in enumerate(hist) if x < beg]
            k = int(rng.choice(idx)) + 1
        else:
            k = k_any
        hist = hist[:k]
    raw = (rng.standard_normal(NV) * 3.0).astype(np.float32)
    # timestamp region: shifted so that the mass rule goes both ways across cases (its sum against the largest text logit)
    raw[beg:] += np.float32(rng.choice([-6.0, -2.0, 0.0, 2.0, 5.0]))
    # a few peaks, some on rule boundaries
    for _ in range(int(rng.integers(0, 4))):
        raw[int(rng.integers(0, NV))] += np.float32(rng.uniform(4, 12))
    if rng.random() < 0.3:
        raw[eot] += np.float32(rng.uniform(4, 12))
    last_ts = max([x for x in hist if x >= beg], default=None)
    if last_ts is not None and rng.random() < 0.5:      # peak right at / below / above the monotonicity boundary
        raw[int(np.clip(last_ts + int(rng.integers(-2, 3)), beg, NV - 1))] += np.float32(rng.uniform(6, 14))
    if kind == "initial" and rng.random() < 0.7:        # peak around max_initial_ts (index 50)
        raw[beg + int(rng.integers(45, 56))] += np.float32(rng.uniform(6, 14))
    if rng.random() < 0.1:
        raw[int(rng.integers(0, NV))] = -np.inf
    return hist, raw


def hf_extra_suppressed(hist, sp, NV):
    """D-a / D-b: ids HF suppresses and whisper.cpp does not, as a function of the history alone."""
    beg = sp["beg"]
    extra = np.zeros(NV, dtype=bool)
    if len(hist) == 0:
        extra[:beg] = True                              # D-a
    else:
        ts = [x for x in hist if x >= beg]
        if ts and hist[-1] < beg:
            extra[ts[-1]] = True                        # D-b (last token is text)
    return extra


def hf_process(hist, raw, sp, NV, static_ids, blank_ids, detect_from_logprob=True):
    """transformers' processors in generate()'s order: SuppressTokens, SuppressTokensAtBegin, WhisperTimeStamp."""
    import torch
    from types import SimpleNamespace
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    cfg = SimpleNamespace(no_timestamps_token_id=sp["not_"], eos_token_id=sp["eot"], bos_token_id=sp["eot"], max_initial_timestamp_index=50,
                          _detect_timestamp_from_logprob=detect_from_logprob)
    ids = torch.tensor([[sp["sot"], sp["sot"] + 1, sp["sot"] + 2][:N_PROMPT] + list(hist)], dtype=torch.long)
    scores = torch.from_numpy(np.asarray(raw, dtype=np.float32)[None, :].copy())
    procs = [SuppressTokensLogitsProcessor(static_ids, device="cpu")]
    if blank_ids:
        procs.append(SuppressTokensAtBeginLogitsProcessor(blank_ids, N_PROMPT, device="cpu"))
    procs.append(WhisperTimeStampLogitsProcessor(cfg, begin_index=N_PROMPT, _detect_timestamp_from_logprob=detect_from_logprob))
    for pr in procs:
        scores = pr(ids, scores)
    s = scores[0]
    logprobs = torch.log_softmax(s.double(), dim=-1)
    return s.numpy(), int(torch.argmax(s).item()), logprobs.numpy()


def mask_hash(filtered):
    return hashlib.sha256(np.packbits(np.isneginf(filtered)).tobytes()).hexdigest()[:16]


KINDS = ["initial", "text", "after_lone_ts", "after_pair", "any"]
