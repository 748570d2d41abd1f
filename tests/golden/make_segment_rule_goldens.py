"""Writes tests/golden/segment_rule_cases.json: seeded one-window token streams (tests/segment_rules_lib.py::make_stream), what the ORACLE's window step made of each, and — checked
at generation time, which needs transformers — that the outcome equals transformers' `_retrieve_segment` on the kinds where whisper.cpp and openai/HF state the same rule, and
equals the documented whisper.cpp behaviour (S-a, S-b, S-e) on the kinds where they differ, with HF's different outcome recorded beside it.
Run from the repository root: python tests/golden/make_segment_rule_goldens.py   (builds nothing; needs oracle/libskw_oracle.so and tools/make_synth_model)"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import logit_rules_lib as lr  # noqa: E402
import segment_rules_lib as sr  # noqa: E402
from oracle_lib import OracleModel  # noqa: E402

N_PER_KIND = 40

if __name__ == "__main__":
    path = "/tmp/skw_segment_rules_tiny.bin"
    subprocess.check_call([os.path.join(ROOT, "tools", "make_synth_model"), path, "--size", "tiny"], stdout=subprocess.DEVNULL)
    om = OracleModel(path); sp = lr.special_ids(om); NV = om.hp.n_vocab; params = om.default_params()
    cases = []
    for ki, kind in enumerate(sr.KINDS):
        for n in range(N_PER_KIND):
            seed = 7000 + 100 * ki + n
            toks, seek, seek_end = sr.make_stream(np.random.default_rng(seed), sp, NV, kind)
            o = sr.oracle_window(om, params, toks, seek, seek_end)
            h = sr.hf_window(toks, sp, seek, seek_end)
            o_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in o["segments"]]
            h_txt = [(a, b, sr.text_ids(t, sp)) for a, b, t in h["segments"] if sr.text_ids(t, sp)]
            want = sr.expected_whisper_cpp(toks, sp, seek, seek_end, kind)
            if kind in sr.AGREE_KINDS:
                assert want is None and not o["failed"] and o["consumed"] == len(toks), (kind, seed)
                assert o_txt == h_txt and o["advance"] == h["advance"], (kind, seed, o, h)
            else:
                assert o["failed"] == want["failed"], (kind, seed, o, want)
                if not want["failed"]:
                    assert o_txt == want["segments"] and o["advance"] == want["advance"], (kind, seed, o, want)
            cases.append(dict(kind=kind, seed=seed, tokens=toks, seek=seek, seek_end=seek_end, failed=o["failed"], consumed=o["consumed"], kept=o["kept"], advance=o["advance"],
                              segments=[[a, b, t] for a, b, t in o["segments"]], hf=dict(advance=h["advance"], segments=[[a, b, t] for a, b, t in h["segments"]])))
    out = dict(what="K12: one window's token-loop bookkeeping and segment assembly; see tests/segment_rules_lib.py", n_vocab=NV, special=sp, cases=cases)
    with open(os.path.join(HERE, "segment_rule_cases.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    n_diff = sum(1 for c in cases if c["kind"] in sr.DIFFER_KINDS)
    print("wrote %d cases (%d on which whisper.cpp and HF differ by rule), %d bytes" % (len(cases), n_diff, os.path.getsize(os.path.join(HERE, "segment_rule_cases.json"))))
