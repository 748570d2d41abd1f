#!/usr/bin/env python3
"""Writes tests/golden/logit_rule_cases.json: seeded K11 cases (decoder history + raw logits from a seed) with the outcome of
whisper_process_logits + whisper_sample_token(best) — admissible-set hash, argmax, log-sum-exp — as oracle/ computes it, every case
checked here against transformers' Whisper logits processors (tests/logit_rules_lib.py: what is compared and which whisper.cpp rules
HF lacks).  Needs transformers + torch (this container); the fixture then travels to the GPU box, where k_dec_sample is run against it.

    python tests/golden/make_logit_rule_goldens.py [n_cases]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import logit_rules_lib as lr  # noqa: E402
from conftest import synth_model  # noqa: E402
from oracle_lib import OracleModel  # noqa: E402


def check_case(om, params, sp, NV, static_ids, blank_ids, seed, kind):
    """-> fixture record; raises AssertionError when oracle and HF disagree."""
    rng = np.random.default_rng(seed)
    hist, raw = lr.make_case(rng, sp, NV, kind)
    extra = lr.hf_extra_suppressed(hist, sp, NV)
    blank = blank_ids if params.suppress_blank else []
    # T1: the index rules alone (timestamp-mass rule off on both sides): HF's suppressed set = whisper.cpp's + the documented extras
    o_idx = lr.oracle_process(om, params, hist, raw, flags=1)
    assert o_idx is not None, "history not producible by the token loop"
    h_idx, _, _ = lr.hf_process(hist, raw, sp, NV, static_ids, blank, detect_from_logprob=False)
    assert np.array_equal(np.isneginf(h_idx), np.isneginf(o_idx[0]) | extra), "index rules differ beyond D-a / D-b (seed %d, %s)" % (seed, kind)
    # T2: all rules, the extras removed from the input of both sides: same admissible set, same mass decision, same argmax, same log-probs
    neutral = raw.copy()
    neutral[extra] = -np.inf
    o_n = lr.oracle_process(om, params, hist, neutral)
    h_n, h_arg, h_lp = lr.hf_process(hist, neutral, sp, NV, static_ids, blank)
    assert np.array_equal(np.isneginf(h_n), np.isneginf(o_n[0])), "admissible sets differ (seed %d, %s)" % (seed, kind)
    assert np.array_equal(h_n[~np.isneginf(h_n)], o_n[0][~np.isneginf(h_n)]), "an admissible logit was changed"
    assert h_arg == o_n[2]["id"], "argmax differs: HF %d, oracle %d (seed %d, %s)" % (h_arg, o_n[2]["id"], seed, kind)
    fin = ~np.isneginf(h_n)
    # D-f: when the mass rule fires whisper.cpp sets the text entries of logits AND logprobs to -inf without renormalising, so a token's plog stays
    # relative to the distribution BEFORE the text was removed: compared with log_softmax of HF's scores with the mass rule left out
    _, _, h_lp_idx = lr.hf_process(hist, neutral, sp, NV, static_ids, blank, detect_from_logprob=False)
    assert np.max(np.abs(h_lp_idx[fin] - o_n[1][fin].astype(np.float64))) < 2e-5, "log-probabilities differ from torch.log_softmax"
    # the un-neutralised input: whisper.cpp's own behaviour where HF's extras matter (oracle only; the index rules above bound the difference)
    o_r = lr.oracle_process(om, params, hist, raw)
    # D-e: no_speech_prob at the first decision (softmax of the unfiltered logits)
    if len(hist) == 0:
        import torch
        ns = float(torch.softmax(torch.from_numpy(raw).double(), -1)[sp["nosp"]])
        assert abs(ns - o_r[3]) <= 1e-9 + 1e-5 * ns, "no_speech_prob differs: %g vs %g" % (ns, o_r[3])
    text_all = bool(np.all(np.isneginf(o_n[0][:sp["beg"]])))
    return {"seed": int(seed), "kind": kind, "hist": [int(x) for x in hist],
            "neutral": {"mask_hash": lr.mask_hash(o_n[0]), "argmax": int(o_n[2]["id"]), "n_admissible": int(fin.sum()), "text_all_suppressed": text_all,
                        "plog": float(o_n[2]["plog"])},
            "raw": {"mask_hash": lr.mask_hash(o_r[0]), "argmax": int(o_r[2]["id"]), "n_admissible": int((~np.isneginf(o_r[0])).sum()), "plog": float(o_r[2]["plog"])}}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    om = OracleModel(synth_model("tiny"))
    NV = om.hp.n_vocab
    sp = lr.special_ids(om)
    blank_ids = lr.rule_ids(om, 2)
    out = {"about": "K11 cases: oracle/ == transformers %s Whisper logits processors on every case (see tests/logit_rules_lib.py); raw logits = "
                    "logit_rules_lib.make_case(numpy default_rng(seed), kind)" % __import__("transformers").__version__,
           "n_vocab": NV, "special": sp, "configs": []}
    for suppress_nst, suppress_blank in ((1, 1), (0, 1), (1, 0)):
        params = om.default_params()
        params.suppress_nst = suppress_nst
        params.suppress_blank = suppress_blank
        static_ids = lr.rule_ids(om, 0) + (lr.rule_ids(om, 1) if suppress_nst else [])
        cases = []
        n = n_cases if (suppress_nst, suppress_blank) == (1, 1) else n_cases // 8
        for k in range(n):
            kind = lr.KINDS[k % len(lr.KINDS)]
            cases.append(check_case(om, params, sp, NV, static_ids, blank_ids, 7000 + 1000 * suppress_nst + 100000 * suppress_blank + k, kind))
        out["configs"].append({"suppress_nst": suppress_nst, "suppress_blank": suppress_blank, "cases": cases})
        forced = sum(c["neutral"]["text_all_suppressed"] for c in cases)
        print("suppress_nst=%d suppress_blank=%d: %d cases agree with transformers; text suppressed (mass rule or grammar) in %d" % (suppress_nst, suppress_blank, n, forced))
    with open(os.path.join(HERE, "logit_rule_cases.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
