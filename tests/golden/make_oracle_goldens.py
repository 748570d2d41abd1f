"""Writes tests/golden/oracle_micro_seed1234.json: the ORACLE's outputs on seeded inputs (micro-sized synthetic model,
tools/make_synth_model --size micro --seed 1234).  These are regression vectors for the oracle and targets for the GPU path;
they are NOT reference outputs (none exist for this path: parity unpinned)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model
from oracle_lib import OracleModel
from streamkit_amd import synth
m = OracleModel(synth_model("micro"))
p = m.default_params(); p.suppress_nst = 1
cases = []
# the last four exercise whisper_full_with_state's delta_min = 10 frames (100 ms): 0.3 s and 0.9 s are transcribed, 30.5 s gets a second
# 0.5 s window, 1500 samples (9 frames) is "too short"
for clip, n in [(0, 480000), (1, 480000), (2, 16000 * 7 + 123), (3, 480768), (4, 16000 * 2), (5, 4800), (6, 14400), (7, 488000), (8, 1500)]:
    r = m.full(synth.clip(clip, n), p)
    cases.append(dict(clip=clip, n_samples=n, tokens=[t[0] for t in r["tokens"]], segments=[[s["t0"], s["t1"], s["text"].decode()] for s in r["segments"]],
                      n_windows=r["n_windows"], fallback_requested=r["fallback_requested"]))
    print(clip, n, len(cases[-1]["tokens"]), r["n_windows"], r["fallback_requested"], "%.3g" % r["min_margin"])
json.dump(dict(model="make_synth_model --size micro --seed 1234", params="defaults + suppress_nst=1", cases=cases), open(os.path.join(HERE, "oracle_micro_seed1234.json"), "w"))
