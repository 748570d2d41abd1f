"""Deterministic synthetic inputs for tests and bench.

Clip c (16 kHz mono f32): x[n] = 0.25 * sum_k a_k sin(2*pi*f_k(c)*n/16000 + phi_k) * env(n) + 0.01*u[n],
f = {180+3c, 700+11c, 2100+17c} Hz, 3-5 Hz amplitude envelope, u uniform(-1,1) from splitmix64(0xC1100000+c).
Inputs are data: they are generated once and handed to every implementation, so float64 numpy math is fine.
"""
import numpy as np

MASK = (1 << 64) - 1


def _splitmix64(seed, n):
    """n outputs of splitmix64 started at `seed` (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def clip(c, n_samples=480000, sample_rate=16000):
    n = np.arange(n_samples, dtype=np.float64)
    t = n / sample_rate
    f = (180.0 + 3.0 * c, 700.0 + 11.0 * c, 2100.0 + 17.0 * c)
    a = (1.0, 0.6, 0.35)
    phi = (0.1 * c, 0.7 + 0.05 * c, 1.9 + 0.03 * c)
    x = np.zeros(n_samples, dtype=np.float64)
    for ak, fk, pk in zip(a, f, phi):
        x += ak * np.sin(2.0 * np.pi * fk * t + pk)
    env = 0.55 + 0.45 * np.sin(2.0 * np.pi * (3.0 + (c % 5) * 0.5) * t + 0.3 * c)
    u = (_splitmix64(0xC1100000 + c, n_samples) >> np.uint64(40)).astype(np.float64) * (1.0 / 8388608.0) - 1.0
    y = 0.25 * x * env / 1.95 * 1.95 + 0.01 * u
    return np.clip(y, -1.0, 1.0).astype(np.float32)


def clips(ids, n_samples=480000):
    return np.stack([clip(c, n_samples) for c in ids])
