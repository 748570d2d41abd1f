"""Independent checker for the polyphase resampler (k_resample_polyphase, include/skw_engine.h skw_resample_polyphase): the documented filter
— Kaiser(beta 8.6)-windowed sinc, cutoff 0.45 x the narrower Nyquist, 32 taps of the slower rate per phase, every phase normalised to unity DC gain —
built here with numpy / scipy.special and applied with scipy.signal.upfirdn in float64.  Test infrastructure only.

How the kernel's indexing maps onto upfirdn.  The kernel computes, for output o with o*M = base*L + ph:
    y[o] = sum_t h[ph][t] * x[base - (T/2 - 1) + t]                                   (zeros outside the signal)
With the prototype P[(j + T/2)*L + ph] = h[ph][T/2 - 1 - j] (j = base - i, the distance in input samples) this is
    y[o] = sum_i P[o*M + (T/2)*L - i*L] * x[i].
upfirdn(P, x3, L, M)[n] = sum_i P[n*M - i*L] * x3[i]; with Zp zeros in front of x (x3[i + Zp] = x[i]) and Zp chosen so that (T/2 + Zp)*L = D*M,
    y[o] = upfirdn(P, x3, L, M)[o + D].
"""
import math

import numpy as np

BETA = 8.6


def design(in_rate, out_rate):
    g = math.gcd(in_rate, out_rate)
    L, M = out_rate // g, in_rate // g
    T = 32 * max(1, (M + L - 1) // L)
    fc = 0.5 * min(1.0, L / M) * 0.90
    from scipy.special import i0
    ph = np.arange(L, dtype=np.float64)[:, None]
    t = np.arange(T, dtype=np.float64)[None, :]
    xpos = (t - (T // 2 - 1)) - ph / L
    r = xpos / (T / 2)
    w = np.where(np.abs(xpos) >= T / 2, 0.0, i0(BETA * np.sqrt(np.clip(1.0 - r * r, 0.0, None))) / i0(BETA))
    h = 2.0 * fc * np.sinc(2.0 * fc * xpos) * w          # np.sinc(x) = sin(pi x) / (pi x)
    h = h / h.sum(axis=1, keepdims=True)
    return L, M, T, h.astype(np.float32)                   # the kernel's table is f32


def reference(x, channels, in_rate, out_rate):
    """x: interleaved f32 [frames * channels] -> interleaved float64 reference output, n_out = ceil(frames * L / M) frames"""
    from scipy.signal import upfirdn
    L, M, T, h = design(in_rate, out_rate)
    P = np.zeros(T * L, dtype=np.float64)
    for ph in range(L):
        for t in range(T):
            j = T // 2 - 1 - t
            P[(j + T // 2) * L + ph] = h[ph, t]
    Zp = (-(T // 2)) % M
    D = (T // 2 + Zp) * L // M
    assert (T // 2 + Zp) * L == D * M
    x = np.asarray(x, dtype=np.float64).reshape(-1, channels)
    n_in = x.shape[0]
    n_out = (n_in * L + M - 1) // M
    out = np.zeros((n_out, channels))
    for c in range(channels):
        x3 = np.concatenate([np.zeros(Zp), x[:, c], np.zeros(T + M)])
        y = upfirdn(P, x3, up=L, down=M)
        out[:, c] = y[D:D + n_out]
    return out.reshape(-1)


def test_signal(seed, frames, channels, in_rate):
    rng = np.random.default_rng(seed)
    n = np.arange(frames)
    cols = []
    for c in range(channels):
        f = rng.uniform(100, 0.45 * in_rate, size=4)
        a = rng.uniform(0.05, 0.25, size=4)
        cols.append(sum(a[k] * np.sin(2 * np.pi * f[k] * n / in_rate + rng.uniform(0, 6.28)) for k in range(4)) + 0.02 * rng.standard_normal(frames))
    return np.stack(cols, axis=1).astype(np.float32).reshape(-1)
