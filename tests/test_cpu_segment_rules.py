"""K12 (one window's token-loop bookkeeping, segment assembly and the advance of seek) pinned against an independent implementation: transformers'
`WhisperGenerationMixin._retrieve_segment` (tests/segment_rules_lib.py lists where whisper.cpp states the step differently, S-a .. S-e, and each difference is asserted as such).
The oracle side is `skwo_debug_window`, which runs the SAME two functions skwo_full runs (token_loop_update, window_output: oracle/skw_oracle.c), so what is pinned here is
what every end-to-end parity test of the GPU path is compared with.  tests/test_gpu_segment_rules.py feeds such streams to the engine (teacher forcing) and compares."""
import importlib.util
import json
import os

import numpy as np
import pytest

import logit_rules_lib as lr
import segment_rules_lib as sr

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "segment_rule_cases.json")


def test_oracle_reproduces_segment_rule_fixture(oracle_tiny):
    fx = json.load(open(FIXTURE))
    om = oracle_tiny
    assert fx["n_vocab"] == om.hp.n_vocab and fx["special"] == lr.special_ids(om)
    params = om.default_params()
    for c in fx["cases"]:
        toks, seek, seek_end = sr.make_stream(np.random.default_rng(c["seed"]), fx["special"], fx["n_vocab"], c["kind"])
        assert (toks, seek, seek_end) == (c["tokens"], c["seek"], c["seek_end"]), c["seed"]            # numpy's generator has not drifted
        o = sr.oracle_window(om, params, toks, seek, seek_end)
        assert o["failed"] == c["failed"], c["seed"]
        if not c["failed"]:
            assert (o["consumed"], o["kept"], o["advance"]) == (c["consumed"], c["kept"], c["advance"]), c["seed"]
            assert [[a, b, t] for a, b, t in o["segments"]] == c["segments"], c["seed"]
    assert len(fx["cases"]) >= 400


def test_fixture_agrees_with_hf_where_the_rules_agree_and_differs_as_documented_where_not():
    """Read off the fixture alone (HF's outcome was recorded beside the oracle's when the fixture was made): no transformers needed."""
    fx = json.load(open(FIXTURE)); sp = fx["special"]
    n = {k: 0 for k in sr.KINDS}
    for c in fx["cases"]:
        o_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in c["segments"]]
        h_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in c["hf"]["segments"] if sr.text_ids(t, sp)]
        k = c["kind"]; n[k] += 1
        if k in sr.AGREE_KINDS:
            assert not c["failed"] and o_txt == h_txt and c["advance"] == c["hf"]["advance"], c["seed"]
        elif k == "unequal_pair":                 # S-a: same ends, same text; a segment starts where the last one ended, HF's at the pair's second member (later)
            assert [s[1:] for s in o_txt] == [s[1:] for s in h_txt] and c["advance"] == c["hf"]["advance"]
            assert all(o_txt[i][0] == o_txt[i - 1][1] for i in range(1, len(o_txt))) and all(h_txt[i][0] > o_txt[i][0] for i in range(1, len(o_txt))), c["seed"]
            assert o_txt[0][0] == h_txt[0][0]
        elif k == "zero_text_last":               # S-b: one segment, same text, same start; whisper.cpp's end and advance are a full chunk, HF's what is left (floored to its 20 ms grid)
            left = min(3000, c["seek_end"] - c["seek"])
            assert len(o_txt) == len(h_txt) == 1 and o_txt[0][0] == h_txt[0][0] and o_txt[0][2] == h_txt[0][2]
            assert o_txt[0][1] == c["seek"] + 3000 and c["seek"] + left - 4 < h_txt[0][1] <= c["seek"] + left and c["advance"] == 3000 and c["hf"]["advance"] == left
        elif k == "zero_text_mid":                # S-e: the pass fails; HF emits the text
            assert c["failed"] and len(h_txt) == 1 and c["hf"]["advance"] == 3000
        elif k == "late_text":                    # S-e: nothing kept but the timestamp, a short hop; HF emits the text and advances a window
            assert not c["failed"] and o_txt == [] and c["kept"] == 1 and c["advance"] == 2 * (c["tokens"][0] - sp["beg"]) > 0 and len(h_txt) == 1 and c["hf"]["advance"] == 3000
        elif k == "empty":
            assert o_txt == [] and h_txt == []
    assert all(v >= 40 for v in n.values()), n


def test_fixture_visits_every_shape():
    fx = json.load(open(FIXTURE)); sp = fx["special"]
    cs = [c for c in fx["cases"] if not c["failed"]]
    assert sum(len(c["segments"]) >= 3 for c in cs) >= 30                                               # several cuts in one window
    assert sum(c["kept"] < c["consumed"] - 1 for c in cs) >= 30                                         # unfinished text thrown away (result_len < tokens sampled)
    assert sum(c["tokens"][-1] != sp["eot"] for c in cs) >= 30                                          # the loop stopped at the end of the audio, not at <|endoftext|>
    assert sum(c["advance"] == min(3000, c["seek_end"] - c["seek"]) < 3000 for c in cs) >= 10           # "text, timestamp" in a short last window
    assert sum(0 < c["advance"] < min(3000, c["seek_end"] - c["seek"]) for c in cs) >= 60               # advance to the last timestamp
    assert sum(c["failed"] for c in fx["cases"]) >= 40


def test_token_loop_failures(oracle_tiny):
    """The two failure rules HF has no counterpart for (S-c), on hand-made streams: a timestamp that steps back; the token budget running out before half a chunk is covered."""
    om = oracle_tiny; sp = lr.special_ids(om); beg, eot = sp["beg"], sp["eot"]; p = om.default_params()
    back = [beg, 11, 12, beg + 200, beg + 200, 13, beg + 150, eot]
    o = sr.oracle_window(om, p, back, 0, 90000)
    assert o["failed"] and o["consumed"] == 7
    n_max = om.hp.n_text_ctx // 2 - 4
    slow = [beg] + [21] * (n_max - 1) + [eot]                                   # no timestamp in n_max tokens
    assert sr.oracle_window(om, p, slow, 0, 90000)["failed"]
    early = [beg, 5, beg + 100, beg + 100] + [21] * (n_max - 4) + [eot]         # a pair at 2.00 s, then text to the budget: 200 frames < half a chunk
    assert sr.oracle_window(om, p, early, 0, 90000)["failed"]
    late = [beg, 5, beg + 800, beg + 800] + [21] * (n_max - 4) + [eot]          # ... a pair at 16.00 s: the pass stands, the text after it goes
    o = sr.oracle_window(om, p, late, 0, 90000)
    assert not o["failed"] and o["advance"] == 1600 and o["kept"] == 4 and len(o["segments"]) == 1


@pytest.mark.skipif(importlib.util.find_spec("transformers") is None, reason="transformers is a tool of the build container")
def test_oracle_agrees_with_transformers_live(oracle_tiny):
    """Fresh seeds (not the fixture's), both sides run now: the agreeing kinds agree, the differing kinds equal the documented whisper.cpp behaviour."""
    om = oracle_tiny; sp = lr.special_ids(om); NV = om.hp.n_vocab; params = om.default_params()
    for ki, kind in enumerate(sr.KINDS):
        for n in range(12):
            toks, seek, seek_end = sr.make_stream(np.random.default_rng(910000 + 100 * ki + n), sp, NV, kind)
            o = sr.oracle_window(om, params, toks, seek, seek_end)
            o_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in o["segments"]]
            want = sr.expected_whisper_cpp(toks, sp, seek, seek_end, kind)
            if want is None:
                h = sr.hf_window(toks, sp, seek, seek_end)
                h_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in h["segments"] if sr.text_ids(t, sp)]
                assert not o["failed"] and o_txt == h_txt and o["advance"] == h["advance"], (kind, n)
            else:
                assert o["failed"] == want["failed"] and (want["failed"] or (o_txt == want["segments"] and o["advance"] == want["advance"])), (kind, n)
