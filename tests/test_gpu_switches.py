"""One parity case per row of the engine's switchboard (streamkit_amd/csrc/skw_engine.hip `g_sw_defs`; skw_kernels.h `SkwSw`): every alternative path a run can select — by the
environment variable SKW_<NAME> at start-up or, as here, in-process through skw_debug_switch_set — is held to the bar of the path it replaces:

  same_bits   the alternative moves data differently and computes the same chains: the exact precision's transcripts (tokens, log-probs, segments) are the default configuration's
              bit for bit — and those are the oracle's — and so are the f16_mfma precision's
  tolerance   the alternative changes a summation order of the f16_mfma precision: exact stays bit-identical, and f16_mfma passes the precision's own bar — every greedy decision
              under teacher forcing against the exact precision (streamkit_amd/parity.py)
  quant       block-quantised files (q5_1): the exact precision's ggml arithmetic / its f16 twin
  resample    the linear resampler's fallback index walks: the default's bits

The table and this file must list the same switches (test_every_switch_has_a_case; tests/test_cpu_abi.py checks the same set without a GPU)."""
import numpy as np
import pytest

from oracle_lib import OracleModel
from streamkit_amd import synth
from streamkit_amd.parity import teacher_forced_compare

pytestmark = pytest.mark.gpu

# (switch, value, kind)
CASES = [
    ("DEC_WFRAG", 0, "same_bits"),
    ("GEMM16W", 0, "same_bits"),
    ("GEMM16W_NGROUPS", 1, "same_bits"), ("GEMM16W_NGROUPS", 2, "same_bits"),
    ("DECODE_GRAPHS", 0, "same_bits"),
    ("DECODE_GROUPS", 1, "same_bits"), ("DECODE_GROUPS", 2, "same_bits"),
    ("DEC_AFRAG", 0, "same_bits"),
    ("XATTN_FRAG", 0, "tolerance"),
    ("DEC_LN_STATS", 0, "tolerance"),
    ("PROMPT_PASS", 0, "tolerance"),
    ("PROMPT_SMALL_GEMM", 1, "tolerance"),
    ("PROMPT_XATTN_MQ", 0, "tolerance"),
    ("DEC_ATTN_FASTV", 0, "tolerance"),
    ("QUANT_TWIN", 1, "quant"),
    ("Q8_LDS", 0, "quant"),
    ("RESAMPLE_SCAN", 1, "resample"),
    ("RESAMPLE_NO_HOST_WALK", 1, "resample"),
]
# Teacher-forced logit-error bound of a tolerance case, where it is not the precision's default for the model (streamkit_amd/parity.py bounds_for: 0.36 for the small test models in rounds 3-4,
# set from batches of <= 1 100 decisions; 0.40 since round 5's random-input hunt, which makes this entry redundant — kept as the record of where 0.363 was first seen).  The two-phase cross attention rounds the NORMALISED probabilities to f16 where the one-pass kernel rounds p <= 1 before the division:
# measured 0.363 over this file's 4 488 decisions (8 argmax disagreements, all at exact-mode margins <= 0.040; gpurun_out/r05a_suite.txt) — bound = 1.1 x that.
TOL_LOGIT_ERR = {("XATTN_FRAG", 0): 0.40}
# a ragged batch: one-window clips, multi-window clips whose later windows carry ~100-token prompts (the prompt pass then has >= 256 rows: the big-tile GEMM and the
# multi-query cross attention run), a clip too short to transcribe
CLIPS = [(11, 16000 * 75), (5, 16000 * 9), (13, 16000 * 47 + 123), (14, 16000 * 93), (15, 16000 * 80), (16, 16000 * 66), (17, 16000 * 88), (8, 1500), (9, 488000), (18, 16000 * 71), (19, 16000 * 30),
         (20, 16000 * 62), (21, 16000 * 3), (22, 16000 * 90), (23, 16000 * 84), (24, 16000 * 77), (25, 16000 * 20), (26, 16000 * 64), (27, 16000 * 92), (28, 16000 * 12)]
MAXS = 16000 * 95


def _key(res):
    return [([tuple(t[:5]) for t in r["tokens"]], [(s["t0"], s["t1"], s["text"]) for s in r["segments"]], r["n_windows"], r["fallback_requested"]) for r in res]


def _transcribe(eng, path, pcms, want_tf=False, quant_mode=1, logit_err_bound=None):
    m = eng.Model(path, quant_mode=quant_mode); ctx = eng.Context(m, max_batch=len(pcms), max_samples=MAXS)
    ctx.set_precision("exact"); ex = _key(ctx.full_batch(pcms))
    ctx.set_precision("f16_mfma"); f16 = _key(ctx.full_batch(pcms))
    tf = teacher_forced_compare(ctx, pcms, logit_err_bound=logit_err_bound) if want_tf else None
    ctx.close(); m.close()
    return ex, f16, tf


@pytest.fixture(scope="module")
def pcms():
    return [synth.clip(c, n) for c, n in CLIPS]


@pytest.fixture(scope="module")
def baseline(eng, tiny_model_path, pcms):
    """the default configuration: its exact transcripts are the oracle's (checked here on a sample; tests/test_gpu_parity.py checks the kernels at large)"""
    assert all(cur == dflt for dflt, cur, _ in eng.switches().values()), "a switch is set in this process's environment: the baseline would not be the default configuration"
    ex, f16, _ = _transcribe(eng, tiny_model_path, pcms)
    om = OracleModel(tiny_model_path)
    for i in (0, 2, 7, 12):
        ro = om.full(pcms[i])
        assert ex[i][0] == [tuple(t[:5]) for t in ro["tokens"]] and ex[i][1] == [(s["t0"], s["t1"], s["text"]) for s in ro["segments"]], CLIPS[i]
    assert sum(k[2] >= 3 for k in ex) >= 8          # enough multi-window clips for >= 256 prompt rows in the later passes
    return ex, f16


def test_every_switch_has_a_case(eng):
    sw = eng.switches()
    assert set(sw) == {c[0] for c in CASES}, (sorted(sw), sorted({c[0] for c in CASES}))
    for name, value, _ in CASES:
        assert value != sw[name][0] or name in ("DECODE_GROUPS", "GEMM16W_NGROUPS"), (name, "the case must select the alternative, not the default")
    with pytest.raises(KeyError):
        with eng.switch("NO_SUCH_SWITCH", 1):
            pass


@pytest.mark.parametrize("name, value, kind", [c for c in CASES if c[2] in ("same_bits", "tolerance")], ids=lambda v: str(v))
def test_transcription_paths(eng, tiny_model_path, pcms, baseline, name, value, kind):
    with eng.switch(name, value):
        ex, f16, tf = _transcribe(eng, tiny_model_path, pcms, want_tf=(kind == "tolerance"), logit_err_bound=TOL_LOGIT_ERR.get((name, value)))
    assert eng.switches()[name][1] == eng.switches()[name][0]            # restored
    for i, (a, b) in enumerate(zip(ex, baseline[0])):
        assert a == b, (name, value, "exact precision", CLIPS[i])
    if kind == "same_bits":
        for i, (a, b) in enumerate(zip(f16, baseline[1])):
            assert a == b, (name, value, "f16_mfma precision", CLIPS[i])
    else:
        print("%s=%d: %d decisions under teacher forcing, %d differ (max exact-mode margin there %s), max logit error %.3g; %d of %d clips identical free-running to the default path"
              % (name, value, tf["steps_checked"], tf["argmax_disagreements"], tf["max_margin_at_disagreement"], tf["max_logit_err"], sum(a == b for a, b in zip(f16, baseline[1])), len(f16)))
        assert tf["ok"], (name, value, tf["max_logit_err"], tf["max_margin_at_disagreement"])


@pytest.mark.parametrize("name, value, kind", [c for c in CASES if c[2] == "quant"], ids=lambda v: str(v))
def test_quantised_file_paths(eng, pcms, name, value, kind):
    from conftest import quantized_model
    path = quantized_model("tiny", "q5_1")
    few = [pcms[1], pcms[10], pcms[12]]
    if name == "QUANT_TWIN":      # the switch = what skw_model_load_ex(..., SKW_QUANT_F16_TWIN) selects through the API
        want = _transcribe(eng, path, few, quant_mode=0)
        with eng.switch(name, value):
            m = eng.Model(path); assert m.quant == 0; m.close()
            got = _transcribe(eng, path, few)
    else:                         # Q8_LDS = 0: the register-fed form of the same integer block dots
        want = _transcribe(eng, path, few)
        with eng.switch(name, value):
            got = _transcribe(eng, path, few)
    assert got[0] == want[0] and got[1] == want[1]
    om = OracleModel(path, quant_mode=0 if name == "QUANT_TWIN" else 1)
    ro = om.full(few[0])
    assert got[0][0][0] == [tuple(t[:5]) for t in ro["tokens"]]


@pytest.mark.parametrize("name, value, kind", [c for c in CASES if c[2] == "resample"], ids=lambda v: str(v))
def test_resampler_index_walks(eng, name, value, kind):
    """44.1 kHz -> 16 kHz (an inexact step: the closed form fails its proof, the proposals are walked) over one long call and several short ones"""
    rng = np.random.default_rng(7)
    chunk, n_chunks = 960, 41
    x = (0.3 * rng.standard_normal(chunk * n_chunks)).astype(np.float32)
    def run():
        dsp = eng.Dsp(0); st = dsp.linear_stream(16000 / 44100, chunk, 1); outs = []; pos = 0; how = []
        for k in (1, 3, n_chunks - 4):
            outs.append(dsp.resample_linear(st, x[pos * chunk:(pos + k) * chunk], k)); how.append(dsp.last_scan_fallback()); pos += k
        dsp.close()
        return np.concatenate(outs), how
    want, how0 = run()
    with eng.switch(name, value):
        got, how1 = run()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if name == "RESAMPLE_SCAN":
        assert how1 == [2, 2, 2] and 2 not in how0      # the single-lane walk ran (and does not by default)
