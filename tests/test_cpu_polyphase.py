"""The polyphase checker itself (tests/polyphase_ref.py): its upfirdn formulation equals the kernel's documented direct form, written out as a
plain loop, and reproduces the committed fixture (tests/golden/polyphase_upfirdn.json) that the GPU kernel is held to."""
import json
import os

import numpy as np
import pytest

import polyphase_ref as pr

HERE = os.path.dirname(os.path.abspath(__file__))


def direct_form(x, ch, in_rate, out_rate):
    """y[o] = sum_t h[ph][t] * x[base - (T/2 - 1) + t], o*M = base*L + ph, zeros outside the signal (skw_kernels.hip k_resample_polyphase)"""
    L, M, T, h = pr.design(in_rate, out_rate)
    x = np.asarray(x, np.float64).reshape(-1, ch)
    n_in = x.shape[0]
    n_out = (n_in * L + M - 1) // M
    y = np.zeros((n_out, ch))
    for o in range(n_out):
        base, ph = divmod(o * M, L)
        for t in range(T):
            i = base - (T // 2 - 1) + t
            if 0 <= i < n_in:
                y[o] += float(h[ph, t]) * x[i]
    return y.reshape(-1)


@pytest.mark.parametrize("in_rate,out_rate,ch", [(48000, 16000, 1), (44100, 16000, 2), (8000, 16000, 1), (22050, 16000, 1)])
def test_upfirdn_formulation_equals_the_direct_form(in_rate, out_rate, ch):
    x = pr.test_signal(5, 700, ch, in_rate)
    a = pr.reference(x, ch, in_rate, out_rate)
    b = direct_form(x, ch, in_rate, out_rate)
    assert a.shape == b.shape and np.max(np.abs(a - b)) < 1e-12


def test_taps_have_the_documented_shape():
    L, M, T, h = pr.design(44100, 16000)
    assert (L, M, T) == (160, 441, 96) and h.shape == (160, 96)
    assert np.allclose(h.sum(axis=1), 1.0, atol=1e-6)                          # unity DC gain per phase
    assert np.argmax(h[0]) == T // 2 - 1                                       # phase 0 peaks on the sample at the output instant
    proto = np.concatenate([h[:, t] for t in range(T - 1, -1, -1)])         # P[(j + T/2) L + ph] = h[ph][T/2 - 1 - j]
    H = np.abs(np.fft.rfft(proto, 1 << 18))
    f = np.arange(H.size) / (1 << 18) * L                                     # cycles per input sample
    assert 20 * np.log10(H[f > 0.5 * 16000 / 44100 * 1.12].max() / H[0]) < -70    # stop band beyond the output Nyquist (+ transition)


def test_fixture_reproduces():
    fx = json.load(open(os.path.join(HERE, "golden", "polyphase_upfirdn.json")))
    for c in fx["cases"]:
        x = pr.test_signal(c["seed"], c["frames"], c["channels"], c["in_rate"])
        y = pr.reference(x, c["channels"], c["in_rate"], c["out_rate"]).reshape(-1, c["channels"])
        assert y.shape[0] == c["n_out"]
        assert np.max(np.abs(y[c["positions"]] - np.array(c["values"]))) < 1e-8
