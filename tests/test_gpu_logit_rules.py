"""K11 on the GPU against the fixture that was checked against transformers' Whisper logits processors (tests/golden/logit_rule_cases.json,
tests/golden/make_logit_rule_goldens.py) and, on the same inputs, against oracle/ bit for bit: both forms of the sampling kernel
(k_dec_sample, register-resident: the decode step's; k_dec_sample_stream: leaves the filtered row in memory, so the admissible SET is compared)."""
import json
import os

import numpy as np
import pytest

import logit_rules_lib as lr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gpu_tiny(tiny_model_path):
    from streamkit_amd import engine
    m = engine.Model(tiny_model_path, device=0)
    c = engine.Context(m, max_batch=64)
    yield c
    c.close(); m.close()


def _batches(cases, n):
    for i in range(0, len(cases), n):
        yield cases[i:i + n]


def test_sampler_matches_transformers_checked_fixture(gpu_tiny, oracle_tiny):
    ctx, om = gpu_tiny, oracle_tiny
    fx = json.load(open(os.path.join(HERE, "golden", "logit_rule_cases.json")))
    sp, NV = fx["special"], fx["n_vocab"]
    assert NV == ctx.model.hp.n_vocab
    n_checked = 0
    for cfg in fx["configs"]:
        p = ctx.default_params(); p.suppress_nst = cfg["suppress_nst"]; p.suppress_blank = cfg["suppress_blank"]
        po = om.default_params(); po.suppress_nst = cfg["suppress_nst"]; po.suppress_blank = cfg["suppress_blank"]
        for batch in _batches(cfg["cases"], 64):
            hists, raws = [], []
            for c in batch:
                h, raw = lr.make_case(np.random.default_rng(c["seed"]), sp, NV, c["kind"])
                assert h == c["hist"]
                hists.append(h); raws.append(raw)
            for variant in ("raw", "neutral"):
                rows = []
                for h, raw in zip(hists, raws):
                    x = raw.copy()
                    if variant == "neutral":
                        x[lr.hf_extra_suppressed(h, sp, NV)] = -np.inf
                    rows.append(x)
                lg = np.stack(rows)
                tk0, tr0, _ = ctx.sample_rows(hists, lg, p, form=0)
                tk1, tr1, filt = ctx.sample_rows(hists, lg, p, form=1, want_filtered=True)
                for r, c in enumerate(batch):
                    want = c[variant]
                    # the admissible set (streaming form) and the decision (both forms) against the HF-checked fixture
                    assert lr.mask_hash(filt[r]) == want["mask_hash"], (c["seed"], variant)
                    assert int(tk0["id"][r]) == want["argmax"] and int(tk1["id"][r]) == want["argmax"], (c["seed"], variant, int(tk0["id"][r]), want["argmax"])
                    assert np.float32(tk0["plog"][r]) == np.float32(want["plog"]) and np.float32(tk1["plog"][r]) == np.float32(want["plog"])
                    # ... and every field of the decision against the oracle on the same input, bit for bit
                    o = lr.oracle_process(om, po, hists[r], rows[r])
                    for f in ("id", "tid", "p", "plog", "pt", "ptsum"):
                        assert np.float32(tk0[f][r]) == np.float32(o[2][f]) and np.float32(tk1[f][r]) == np.float32(o[2][f]), (c["seed"], variant, f, tk0[f][r], tk1[f][r], o[2][f])
                    assert np.array_equal(np.isneginf(filt[r]), np.isneginf(o[0]))
                    fin = ~np.isneginf(o[0])
                    assert np.array_equal(filt[r][fin], o[0][fin])
                    # the trace record: the two largest admissible logits and their owners (what teacher forcing compares)
                    srt = np.sort(o[0][fin])[::-1]
                    assert tr0["top1"][r] == srt[0] and int(tr0["top1_id"][r]) == int(np.flatnonzero(o[0] == srt[0])[0])
                    if srt.size > 1:
                        assert tr0["top2"][r] == srt[1]
                    n_checked += 1
    assert n_checked >= 2 * 300


def test_sampler_rejects_a_history_the_token_loop_cannot_produce(gpu_tiny):
    ctx = gpu_tiny
    fx = json.load(open(os.path.join(HERE, "golden", "logit_rule_cases.json")))
    beg = fx["special"]["beg"]
    with pytest.raises(RuntimeError, match="cannot produce this history"):
        ctx.sample_rows([[beg + 100, 11, beg + 50]], np.zeros((1, fx["n_vocab"]), np.float32))
