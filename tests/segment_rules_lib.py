"""K12 (segment assembly and the advance of `seek`, whisper_full_with_state after the token loop) on its own: seeded token streams, the oracle's side, and an INDEPENDENT
implementation of the same step — transformers' `WhisperGenerationMixin._retrieve_segment`, HF's restatement of openai/whisper `transcribe()` (the loop whisper.cpp's own follows).
Test infrastructure only; transformers is a tool of the build container (the committed fixture tests/golden/segment_rule_cases.json carries the checked outcomes).

What the two sides are given: ONE window's sampled token ids in order (timestamps mode, the stream the token loop receives; HF's `seek_sequence` is the same ids without the
final <|endoftext|>), the window's position `seek` and the end of the audio `seek_end` in 10 ms frames.  What is compared: the segments' (t0, t1) in centiseconds, each segment's
TEXT token ids (ids below <|endoftext|>: whisper.cpp keeps the closing timestamp run's first member with a segment, HF keeps both members of a pair), and the advance of seek.

Where whisper.cpp (as recalled) and openai/HF state the step differently — each is generated on purpose and asserted as a difference, never papered over:
  S-a  a timestamp pair whose members differ (<|5.00|><|5.20|>): whisper.cpp starts the next segment where the last one ended (t0 = t1 = 5.00); HF starts it at the pair's second
       member (5.20).  Ends agree.  whisper.cpp then advances seek by the LAST timestamp seen (5.20 if it closes the stream), HF by the pair's first member.
  S-b  the LAST window of the audio, tokens "<|0.00|>, text, <|endoftext|>" (no timestamp above <|0.00|>): whisper.cpp ends the segment at seek + 3000 and advances by 3000
       (seek_delta keeps its initial value, a full chunk) whatever is left of the audio; HF ends it at, and advances by, the frames that are left.  Equal when 3000 are left.
  S-c  whisper.cpp's token loop stops by itself when a timestamp reaches the end of the audio (seek + 2 k + 10 >= seek_end) and FAILS the pass when a timestamp steps back or
       the token budget runs out before half a chunk is covered; HF has no such step (its generate() runs to <|endoftext|>).  Streams here end where whisper.cpp's loop ends.
  S-d  <|0.00|> (id == token_beg) is a timestamp for HF (`ge`) and not for whisper.cpp's cut rule (`> token_beg`); the grammar only produces it as a window's first token,
       where the two readings coincide.
  S-e  whisper.cpp keeps a window's tokens only up to the LAST timestamp above <|0.00|> (result_len) and advances to it.  openai/HF do the same once a pair has closed, but
       with no closed pair they keep everything and advance a full window.  So "<|k|>, text, <|endoftext|>" with k > 0 yields NO segment and a hop of 2 k frames in whisper.cpp
       (the text is decoded again from there) against one segment and a full advance in HF; and "<|0.00|>, text, <|endoftext|>" with audio left beyond the window FAILS the
       pass in whisper.cpp ("result_len = 0": the temperature ladder takes over) where HF emits the text.  "<|k|>, <|endoftext|>" alike.
Everything else — cuts at pairs, the trailing unfinished segment thrown away, the full-window advance after "text, timestamp" — must agree exactly."""
import ctypes as C

import numpy as np

from oracle_lib import lib, Params

AGREE_KINDS = ("pairs_then_single", "pairs_then_text", "single_only", "reaches_end", "short_window")
DIFFER_KINDS = ("unequal_pair", "zero_text_last", "zero_text_mid", "late_text", "empty")
KINDS = AGREE_KINDS + DIFFER_KINDS


def bind():
    L = lib()
    L.skwo_debug_window.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.skwo_debug_window.restype = C.c_int
    return L


def oracle_window(om, params, toks, seek, seek_end):
    """-> dict(failed, consumed, kept, advance, segments=[(t0, t1, [ids])])"""
    L = bind()
    t = np.ascontiguousarray(toks, dtype=np.int32)
    seg_t = np.zeros(2 * 64, np.int64); seg_off = np.zeros(65, np.int32); seg_tok = np.zeros(max(1, t.size), np.int32)
    n_seg, adv, kept, cons, failed = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = L.skwo_debug_window(om.h, C.byref(params), t.ctypes.data, t.size, seek, seek_end, seg_t.ctypes.data, seg_off.ctypes.data, seg_tok.ctypes.data, 64,
                             C.byref(n_seg), C.byref(adv), C.byref(kept), C.byref(cons), C.byref(failed))
    assert rc == 0
    segs = [(int(seg_t[2 * i]), int(seg_t[2 * i + 1]), [int(x) for x in seg_tok[seg_off[i]:seg_off[i + 1]]]) for i in range(n_seg.value)]
    return dict(failed=bool(failed.value), consumed=cons.value, kept=kept.value, advance=adv.value, segments=segs)


def make_stream(rng, sp, NV, kind, whole_clip=False):
    """One window's sampled ids as the timestamp grammar allows them (first a timestamp <= 1.00 s; text; "text, ts, ts" pairs; "text, ts" only before <|endoftext|>; timestamps
    never decrease), shaped by `kind`.  Returns (tokens incl. the final <|endoftext|> where the loop needs one, seek, seek_end).  whole_clip: the window is the first AND last
    of a clip (seek 0, seek_end = the frames it has), the form tests/test_gpu_segment_rules.py can hand to the engine as audio of that length."""
    beg, eot = sp["beg"], sp["eot"]
    text = lambda: [int(x) for x in rng.integers(0, min(eot, 2000), size=int(rng.integers(1, 9)))]
    seek = int(rng.choice([0, 0, 3000, 2912, 45000]))
    last = kind in ("short_window", "reaches_end", "zero_text_last")                     # the audio ends inside this window
    frames_left = int(rng.integers(150, 2900)) if kind == "short_window" or (kind == "zero_text_last" and rng.random() < 0.7) else 3000
    seek_end = seek + frames_left + (0 if last else int(rng.integers(100, 90000)))
    if whole_clip:
        seek, seek_end = 0, frames_left
    k_hi = frames_left // 2 - 6                      # timestamps stay clear of the end of the audio unless the kind asks for it (S-c)
    k = 0 if kind in ("single_only", "zero_text_last", "zero_text_mid") else int(rng.integers(1, 51)) if kind == "late_text" else int(rng.integers(0, 51))
    toks = [beg + k]
    if kind == "empty":
        return toks + [eot], seek, seek_end
    if kind in ("zero_text_last", "zero_text_mid", "late_text"):
        return toks + text() + [eot], seek, seek_end
    n_pairs = 0 if kind == "single_only" else int(rng.integers(1, 5))
    for _ in range(n_pairs):
        toks += text()
        k = min(k + int(rng.integers(1, 400)), k_hi)
        k2 = min(k + int(rng.integers(1, 20)), k_hi) if kind == "unequal_pair" else k
        toks += [beg + k, beg + k2]
        k = k2
    if kind in ("pairs_then_single", "single_only", "unequal_pair") or (kind == "short_window" and rng.random() < 0.5):
        toks += text()
        k = min(k + int(rng.integers(1, 400)), k_hi)
        toks += [beg + k, eot]                       # "text, timestamp" and nothing after it
    elif kind == "reaches_end":
        toks += text()
        toks += [beg + (frames_left - 10 + 1) // 2 + int(rng.integers(0, 3))]      # 2 k + 10 >= frames left: the loop stops here, no <|endoftext|>
    else:                                            # pairs_then_text, the other half of short_window: unfinished text, then <|endoftext|>
        toks += text() + [eot]
    return toks, seek, seek_end


def hf_window(toks, sp, seek, seek_end):
    """transformers' `_retrieve_segment` on the same stream (without the final <|endoftext|>, which generate() strips) -> dict(advance, segments=[(t0, t1, [ids])])"""
    import torch
    from transformers.models.whisper.generation_whisper import WhisperGenerationMixin
    ids = [t for t in toks if t != sp["eot"]]
    seq = torch.tensor(ids, dtype=torch.long)
    frames = min(3000, seek_end - seek)
    segs, off = WhisperGenerationMixin._retrieve_segment(
        seek_sequence=seq, seek_outputs=[None], time_offset=torch.tensor([seek * 0.01], dtype=torch.float64), timestamp_begin=sp["beg"],
        seek_num_frames=torch.tensor([frames]), time_precision=0.02, time_precision_features=0.01, input_stride=2, prev_idx=0, idx=0,
        return_token_timestamps=False, decoder_input_ids=torch.zeros((1, 3), dtype=torch.long))
    out = [(int(round(float(s["start"]) * 100)), int(round(float(s["end"]) * 100)), [int(x) for x in s["tokens"].tolist()]) for s in segs]
    return dict(advance=int(off), segments=out)


def text_ids(ids, sp):
    return [t for t in ids if t < sp["eot"]]


def expected_whisper_cpp(toks, sp, seek, seek_end, kind):
    """What S-a / S-b / S-e say whisper.cpp does on the kinds where it differs from HF, derived from the stream alone (a third statement of the rule, in prose-sized Python):
    -> dict(failed, advance, segments=[(t0, t1, [text ids])]) or None for the kinds that simply agree with HF."""
    beg, eot = sp["beg"], sp["eot"]
    k0 = toks[0] - beg
    body = [t for t in toks[1:] if t < eot]
    if kind in ("zero_text_mid",) or (kind == "empty" and k0 == 0 and seek + 3010 < seek_end):
        return dict(failed=True, advance=0, segments=[])
    if kind == "empty" and k0 == 0:
        return dict(failed=False, advance=3000, segments=[])
    if kind in ("late_text", "empty"):
        return dict(failed=False, advance=2 * k0, segments=[])
    if kind == "zero_text_last":
        return dict(failed=False, advance=3000, segments=[(seek, seek + 3000, body)])
    if kind == "unequal_pair":        # cut at every timestamp run; a segment starts where the last one ended; "text, timestamp" at the end: a full advance
        segs, t0, cur, i = [], seek + 2 * k0, [], 1
        while i < len(toks) and toks[i] != eot:
            if toks[i] > beg:
                t1 = seek + 2 * (toks[i] - beg)
                if cur:
                    segs.append((t0, t1, cur))
                cur, t0 = [], t1
                while i < len(toks) and toks[i] > beg:
                    i += 1
                continue
            cur.append(toks[i]); i += 1
        return dict(failed=False, advance=min(3000, seek_end - seek), segments=segs)
    return None
