"""K11 (whisper_process_logits + greedy sampling) pinned against an independent implementation of the same rules: transformers'
Whisper logits processors (tests/logit_rules_lib.py lists what whisper.cpp has that HF lacks, D-a .. D-f).  The committed fixture
tests/golden/logit_rule_cases.json carries the checked outcomes to the GPU box (tests/test_gpu_logit_rules.py runs k_dec_sample on it)."""
import importlib.util
import json
import os

import numpy as np
import pytest

import logit_rules_lib as lr

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "logit_rule_cases.json")


def _params(om, cfg):
    p = om.default_params()
    p.suppress_nst = cfg["suppress_nst"]
    p.suppress_blank = cfg["suppress_blank"]
    return p


def test_oracle_reproduces_logit_rule_fixture(oracle_tiny):
    """every fixture case: same history from the seed (numpy's generator has not drifted), same admissible set, same argmax, same plog"""
    fx = json.load(open(FIXTURE))
    om = oracle_tiny
    assert fx["n_vocab"] == om.hp.n_vocab and fx["special"] == lr.special_ids(om)
    n = 0
    for cfg in fx["configs"]:
        params = _params(om, cfg)
        for c in cfg["cases"]:
            hist, raw = lr.make_case(np.random.default_rng(c["seed"]), fx["special"], fx["n_vocab"], c["kind"])
            assert hist == c["hist"]
            neutral = raw.copy()
            neutral[lr.hf_extra_suppressed(hist, fx["special"], fx["n_vocab"])] = -np.inf
            for name, x in (("raw", raw), ("neutral", neutral)):
                o = lr.oracle_process(om, params, hist, x)
                assert lr.mask_hash(o[0]) == c[name]["mask_hash"], (c["seed"], name)
                assert o[2]["id"] == c[name]["argmax"], (c["seed"], name)
                assert int((~np.isneginf(o[0])).sum()) == c[name]["n_admissible"]
                assert np.float32(o[2]["plog"]) == np.float32(c[name]["plog"])
            n += 1
    assert n >= 300


def test_fixture_visits_every_rule():
    """the cases reach each branch of the rules: first decision, after a lone timestamp, after a pair, in text; the mass rule both ways; EOT chosen"""
    fx = json.load(open(FIXTURE))
    sp = fx["special"]
    cases = [c for cfg in fx["configs"] for c in cfg["cases"]]
    kinds = {k: sum(c["kind"] == k for c in cases) for k in lr.KINDS}
    assert all(v >= 40 for v in kinds.values()), kinds
    in_text = [c for c in cases if c["hist"] and c["hist"][-1] < sp["beg"]]
    assert sum(c["neutral"]["text_all_suppressed"] for c in in_text) >= 10          # the mass rule fired
    assert sum(not c["neutral"]["text_all_suppressed"] for c in in_text) >= 10      # ... and did not
    assert sum(c["raw"]["argmax"] == sp["eot"] for c in cases) >= 5
    assert sum(c["raw"]["argmax"] != c["neutral"]["argmax"] for c in cases) >= 5    # D-a / D-b change outcomes: whisper.cpp's own behaviour is pinned too
    assert any(c["hist"][:1] == [sp["beg"]] for c in cases if c["hist"])            # a lone <|0.00|>: has_ts stays 0


@pytest.mark.skipif(importlib.util.find_spec("transformers") is None, reason="transformers is a tool of the build container")
def test_oracle_agrees_with_transformers_processors(oracle_tiny):
    """fresh seeds (not the fixture's) through the generator script's own check: index rules = HF's minus D-a / D-b, and with those removed from
    the input the admissible set, the timestamp-mass decision, the argmax and the log-probabilities all agree"""
    spec = importlib.util.spec_from_file_location("make_logit_rule_goldens", os.path.join(HERE, "golden", "make_logit_rule_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    om = oracle_tiny
    NV, sp = om.hp.n_vocab, lr.special_ids(om)
    blank = lr.rule_ids(om, 2)
    for suppress_nst in (1, 0):
        params = om.default_params()
        params.suppress_nst = suppress_nst
        static_ids = lr.rule_ids(om, 0) + (lr.rule_ids(om, 1) if suppress_nst else [])
        for k in range(30):
            mod.check_case(om, params, sp, NV, static_ids, blank, 990000 + 1000 * suppress_nst + k, lr.KINDS[k % len(lr.KINDS)])


def test_history_the_token_loop_cannot_produce_is_rejected(oracle_tiny):
    om = oracle_tiny
    sp = lr.special_ids(om)
    raw = np.zeros(om.hp.n_vocab, dtype=np.float32)
    # a timestamp that goes backwards after text: whisper_full_with_state marks the decoder failed there
    assert lr.oracle_process(om, om.default_params(), [sp["beg"] + 100, 11, sp["beg"] + 50], raw) is None
