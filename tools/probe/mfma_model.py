#!/usr/bin/env python3
"""What does v_mfma_f32_16x16x32_f16 compute, exactly?  (round 5; DESIGN.md section 6 "what comes next": a CPU-restatable f16 matrix-core contraction would let the oracle check the
f16_mfma precision bit for bit.)  Generates operand sets for tools/probe/probe_mfma_run, and fits a parametrised model of the hardware's arithmetic to the results with exact integer
arithmetic.   usage: mfma_model.py gen DIR    |    mfma_model.py fit DIR"""
import sys, os, itertools
import numpy as np
from fractions import Fraction

P = 48                                       # problems per set (256 outputs each)
SETS = ["g0", "g3", "g0_noacc", "all32"]     # products only in k-group 0 / only in group 3 / group 0 with C = 0 / all 32 k's


def rnd_f16(rng, n, emin, emax):
    e = rng.integers(emin, emax + 1, n); m = rng.integers(0, 1024, n); s = rng.integers(0, 2, n) * 2 - 1
    return (s * (1.0 + m / 1024.0) * np.exp2(e.astype(np.float64))).astype(np.float16)


def gen(d):
    rng = np.random.default_rng(20251005)
    for name in SETS:
        A = np.zeros((P, 16, 32), np.float16); B = np.zeros((P, 32, 16), np.float16)
        ks = {"g0": range(0, 8), "g3": range(24, 32), "g0_noacc": range(0, 8), "all32": range(32)}[name]
        for p in range(P):
            spread = [2, 4, 6][p % 3]                                   # exponent spread of the operands: products differ by up to 4 x spread binades
            for k in ks:
                A[p, :, k] = rnd_f16(rng, 16, -spread, spread); B[p, k, :] = rnd_f16(rng, 16, -spread, spread)
        C = np.zeros((P, 16, 16), np.float32)
        if name != "g0_noacc":
            e = rng.integers(-12, 14, (P, 16, 16)); m = rng.integers(0, 1 << 23, (P, 16, 16)); s = rng.integers(0, 2, (P, 16, 16)) * 2 - 1
            C = (s * (1.0 + m / float(1 << 23)) * np.exp2(e.astype(np.float64))).astype(np.float32)
            C[rng.random((P, 16, 16)) < 0.1] = 0.0
        A.tofile(os.path.join(d, name + "_A.bin")); B.tofile(os.path.join(d, name + "_B.bin")); C.tofile(os.path.join(d, name + "_C.bin"))
    print("wrote", SETS, "x", P, "problems to", d)


def frexp_int(x):
    """float (python) -> (integer mantissa, exponent) exactly, mantissa odd or zero"""
    if x == 0.0:
        return 0, 0
    f = Fraction(x); n, dnm = f.numerator, f.denominator      # dnm is a power of two
    e = -(dnm.bit_length() - 1)
    while n % 2 == 0:
        n //= 2; e += 1
    return n, e


def lead(n, e):      # exponent of the leading bit of n * 2^e
    return abs(n).bit_length() - 1 + e


def shift_round(n, sh, mode):
    """n / 2^sh as an integer: mode 'rz' toward zero, 'fl' toward -inf (arithmetic shift), 'rne' nearest-even"""
    if sh <= 0:
        return n << (-sh)
    if mode == "fl":
        return n >> sh
    if mode == "rz":
        return -((-n) >> sh) if n < 0 else n >> sh
    q, r = divmod(n, 1 << sh); half = 1 << (sh - 1)
    if r > half or (r == half and (q & 1)):
        q += 1
    return q


def to_f32(n, e, mode):
    """n * 2^e rounded to float32 (mode 'rne' / 'rz'); assumes the normal range"""
    if n == 0:
        return np.float32(0.0)
    s = -1 if n < 0 else 1; n = abs(n); bl = n.bit_length()
    if bl > 24:
        sh = bl - 24
        if mode == "rz":
            n >>= sh
        else:
            q, r = divmod(n, 1 << sh); half = 1 << (sh - 1)
            if r > half or (r == half and (q & 1)):
                q += 1
            n = q
        e += sh
    return np.float32(s * float(n) * 2.0 ** e) if abs(e) < 1000 else np.float32(s * np.inf)


def fused_add(terms, W, tmode, fmode, unnorm_exps=None):
    """terms: list of (n, e) exact values (acc first); align every term to the largest leading-bit exponent, keep W bits below it (tmode per term), sum, round to f32 (fmode)"""
    nz = [(n, e) for n, e in terms if n]
    if not nz:
        return 0, 0
    if unnorm_exps is None:
        emax = max(lead(n, e) for n, e in nz)
    else:
        emax = max(x for x, (n, e) in zip(unnorm_exps, terms) if n)
    lsb = emax - W
    S = sum(shift_round(n, lsb - e, tmode) for n, e in nz)
    return S, lsb


def fit(d):
    for name in SETS:
        A = np.fromfile(os.path.join(d, name + "_A.bin"), np.float16).reshape(P, 16, 32); B = np.fromfile(os.path.join(d, name + "_B.bin"), np.float16).reshape(P, 32, 16)
        C = np.fromfile(os.path.join(d, name + "_C.bin"), np.float32).reshape(P, 16, 16); D = np.fromfile(os.path.join(d, name + "_D.bin"), np.float32).reshape(P, 16, 16)
        cases = []
        for p in range(0, P, 2):
            for i in range(16):
                for j in range(0, 16, 3):
                    prods = []
                    for k in range(32):
                        a, b = float(A[p, i, k]), float(B[p, k, j])
                        na, ea = frexp_int(a); nb, eb = frexp_int(b)
                        ua = (int(np.floor(np.log2(abs(a)))) if a else 0) + (int(np.floor(np.log2(abs(b)))) if b else 0)      # unnormalised product exponent ea + eb
                        prods.append((na * nb, ea + eb, ua))
                    cases.append((frexp_int(float(C[p, i, j])), prods, D[p, i, j]))
        print("== set %s: %d cases" % (name, len(cases)))
        best = []
        for W in range(22, 34):
            for tmode in ("rz", "fl", "rne"):
                for fmode in ("rne", "rz"):
                    for un in (False, True):
                        bad = 0
                        for (c, prods, dref) in cases:
                            acc = c
                            for g in range(4):
                                grp = prods[8 * g:8 * g + 8]
                                if not any(n for n, _, _ in grp):
                                    continue
                                terms = [acc] + [(n, e) for n, e, _ in grp]
                                ue = None
                                if un:
                                    ue = [lead(*acc) if acc[0] else 0] + [u for _, _, u in grp]
                                S, lsb = fused_add(terms, W, tmode, fmode, ue)
                                v = to_f32(S, lsb, fmode)
                                acc = frexp_int(float(v))
                            got = np.float32(float(Fraction(acc[0]) * Fraction(2) ** acc[1])) if acc[0] else np.float32(0.0)
                            bad += int(got != dref and not (got == 0 and dref == 0))
                        best.append((bad, W, tmode, fmode, un))
        best.sort()
        for b in best[:8]:
            print("   mismatches %5d   W=%d term=%s final=%s unnormalised_product_exponent=%s" % b)


if __name__ == "__main__":
    (gen if sys.argv[1] == "gen" else fit)(sys.argv[2])
