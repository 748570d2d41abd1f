"""K12 on the GPU path: the engine's own token-loop bookkeeping (k_dec_sample's tail, on the device) and segment assembly (host side of skw_full_batch) are fed seeded token
streams by teacher forcing — `skw_full_batch_traced` with forced ids: the decoder consumes exactly these tokens, whatever its logits say — on audio of exactly the length the
stream's window assumes, and must cut the same segments with the same times and text, and run the same number of windows, as the oracle's window step, which
tests/test_cpu_segment_rules.py holds against transformers' `_retrieve_segment` and against the documented whisper.cpp differences.
Second windows: where the first window advances by less than the clip (it ended in unfinished text, S-e's short hop), the stream continues with "<|0.00|>, <|endoftext|>",
which the last window of a clip accepts and turns into no segment."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logit_rules_lib as lr  # noqa: E402
import segment_rules_lib as sr  # noqa: E402

pytestmark = pytest.mark.gpu
KINDS = ("pairs_then_single", "pairs_then_text", "single_only", "reaches_end", "short_window", "unequal_pair", "zero_text_last", "late_text")


def test_engine_cuts_forced_streams_like_the_oracle(tiny_model_path, oracle_tiny):
    from streamkit_amd import engine, synth
    om = oracle_tiny; sp = lr.special_ids(om); NV = om.hp.n_vocab; beg, eot = sp["beg"], sp["eot"]
    gm = engine.Model(tiny_model_path); ctx = engine.Context(gm, max_batch=16, max_samples=480000)
    p = ctx.default_params(); p.temperature_inc = 0.0; p.logprob_thold = -1.0e30; p.entropy_thold = -1.0e30      # one pass per window whatever the forced tokens' probabilities are
    op = om.default_params(); op.temperature_inc = 0.0
    cases = []
    for ki, kind in enumerate(KINDS):
        for n in range(8):
            toks, seek, seek_end = sr.make_stream(np.random.default_rng(555000 + 100 * ki + n), sp, NV, kind, whole_clip=True)
            o = sr.oracle_window(om, op, toks, seek, seek_end)
            assert not o["failed"] and o["consumed"] == len(toks), (kind, n)
            forced, windows = list(toks), 1
            if o["advance"] + 10 < seek_end:           # the clip goes on: one more (last) window
                forced += [beg, eot]; windows = 2
                o2 = sr.oracle_window(om, op, [beg, eot], o["advance"], seek_end)
                assert not o2["failed"] and o2["segments"] == [] and o["advance"] + o2["advance"] + 10 >= seek_end
            cases.append((kind, n, toks, seek_end, forced, windows, o))
    n_checked = 0
    for c0 in range(0, len(cases), 16):
        chunk = cases[c0:c0 + 16]
        clips = [synth.clip(i, n_samples=c[3] * 160) for i, c in enumerate(chunk)]
        res, traces = ctx.full_batch(clips, params=p, forced=[np.asarray(c[4], np.int32) for c in chunk])
        for (kind, n, toks, seek_end, forced, windows, o), r, tr in zip(chunk, res, traces):
            assert len(tr) == len(forced) and [int(x) for x in tr["forced_id"]] == forced, (kind, n)              # every forced token was consumed, none left over
            assert r["n_windows"] == windows, (kind, n, r["n_windows"], windows)
            got = [(s["t0"], s["t1"], bytes(s["text"])) for s in r["segments"]]
            want = [(a, b, b"".join(om.token_bytes(t) for t in sr.text_ids(ids, sp))) for a, b, ids in o["segments"]]
            assert got == want, (kind, n, got, want)
            n_checked += 1
    assert n_checked == len(KINDS) * 8
    ctx.close(); gm.close()
