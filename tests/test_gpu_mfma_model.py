"""The f16_mfma precision's contractions, reproduced on the CPU bit for bit.

Rounds 1-4 checked f16_mfma only through tolerances ("the matrix cores' summation order no CPU loop restates").  include/skw_mfma_model.h now restates the instruction
(`v_mfma_f32_16x16x32_f16`) in integer arithmetic and oracle/'s skwo_gemm_f16mfma chains it the way the kernels do; here the GPU's own kernels are held to that:
  * single instructions on this box against the committed MI355X vectors (the fixture cannot go stale) and against the model on fresh seeded operands;
  * the encoder's two GEMM kernels (k_gemm16w, k_gemm16), the decode step's K-split kernel (k_gemm16_small) and the vocabulary kernel (k_gemm16_vocab), plain f32 output,
    at the model sizes' own shapes: EVERY output element bit-identical to the oracle's.
What this pins: the summation order of each kernel (operand slots in memory order, k-blocks ascending, the decode kernel's four K quarters added ((q0 + q1) + q2) + q3) — an edit
that reorders MFMAs per accumulator, or a compiler that reassociates the partial sums, changes bits here.  No reference file is involved (the precision is this repository's own)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfma_f16_hw_vectors.npz")
SETS = ["g0", "g3", "g0_noacc", "all32", "accdom", "cancel", "sub", "window", "w21", "w22", "w23", "w24", "w25", "w26"]
KERNEL = {"k_gemm16w": 0, "k_gemm16": 1, "decode": 2}


@pytest.fixture(scope="module")
def ctx(tiny_model_path):
    from streamkit_amd import engine
    m = engine.Model(tiny_model_path); c = engine.Context(m, max_batch=1, max_samples=16000)
    c.set_precision("f16_mfma")
    L = engine.lib()
    L.skw_debug_mfma16x32.restype = C.c_int
    L.skw_debug_mfma16x32.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.skw_debug_gemm16_out.restype = C.c_int
    L.skw_debug_gemm16_out.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    yield c, L
    c.close(); m.close()


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def _hw_tiles(ctx, A, B, Cc):
    c, L = ctx
    A = np.ascontiguousarray(A, np.uint16); B = np.ascontiguousarray(B, np.uint16); Cc = np.ascontiguousarray(Cc, np.float32)
    D = np.empty_like(Cc)
    assert L.skw_debug_mfma16x32(c.h, A.shape[0], A.ctypes.data, B.ctypes.data, Cc.ctypes.data, D.ctypes.data) == 0, c.last_error()
    return D


def _hw_gemm(ctx, kernel, A, W):
    c, L = ctx
    M, K = A.shape; N = W.shape[0]
    out = np.empty((M, N), np.float32)
    assert L.skw_debug_gemm16_out(c.h, KERNEL[kernel], M, N, K, A.ctypes.data, W.ctypes.data, out.ctypes.data) == 0, c.last_error()
    return out


@pytest.mark.parametrize("name", SETS)
def test_committed_vectors_are_what_this_gpu_returns(ctx, name):
    z = np.load(GOLD)
    D = _hw_tiles(ctx, z[name + "_A"], z[name + "_B"], z[name + "_C"])
    assert np.array_equal(_bits(D), _bits(z[name + "_D"]))


def _operands(rng, P, kind):
    """f16 operand tiles with the exponent spreads a GEMM on this path meets and the ones that separate models: `act` = activations x weights, `wide` = exponents all over
    the f16 range (alignment losses everywhere), `tie` = few-bit mantissas (sums that land on rounding ties), `cross` = accumulators next to powers of two, `tiny` = subnormal-heavy."""
    if kind == "act":
        a = rng.standard_normal((P, 16, 32)) * np.exp2(rng.integers(-2, 5, (P, 16, 1))); b = rng.standard_normal((P, 32, 16)) * 0.05; c = rng.standard_normal((P, 16, 16)) * 4
    elif kind == "wide":
        a = rng.standard_normal((P, 16, 32)) * np.exp2(rng.integers(-12, 8, (P, 16, 32)).astype(np.float64))
        b = rng.standard_normal((P, 32, 16)) * np.exp2(rng.integers(-12, 6, (P, 32, 16)).astype(np.float64))
        c = rng.standard_normal((P, 16, 16)) * np.exp2(rng.integers(-20, 16, (P, 16, 16)).astype(np.float64))
    elif kind == "tie":
        a = rng.integers(-3, 4, (P, 16, 32)) * np.exp2(rng.integers(-10, 10, (P, 16, 32)).astype(np.float64))
        b = rng.integers(-3, 4, (P, 32, 16)) * np.exp2(rng.integers(-10, 4, (P, 32, 16)).astype(np.float64))
        c = rng.integers(-5, 6, (P, 16, 16)) * np.exp2(rng.integers(-24, 12, (P, 16, 16)).astype(np.float64))
    elif kind == "cross":        # running sums that cross a power of two between the instruction's four additions
        a = rng.standard_normal((P, 16, 32)) * np.exp2(rng.integers(-2, 3, (P, 16, 1)).astype(np.float64))
        b = rng.standard_normal((P, 32, 16)) * np.exp2(rng.integers(-9, -3, (P, 1, 16)).astype(np.float64))
        c = np.exp2(rng.integers(-3, 12, (P, 16, 16)).astype(np.float64)) * (1 + rng.standard_normal((P, 16, 16)) * 2.0 ** rng.integers(-12, -3, (P, 16, 16))) * rng.choice([-1.0, 1.0], (P, 16, 16))
    else:
        a = rng.standard_normal((P, 16, 32)) * 2.0 ** -15; b = rng.standard_normal((P, 32, 16)) * np.exp2(rng.integers(-3, 12, (P, 32, 16)).astype(np.float64))
        c = rng.standard_normal((P, 16, 16)) * 2.0 ** -12
    with np.errstate(over="ignore"):
        a16 = np.clip(a, -60000, 60000).astype(np.float16); b16 = np.clip(b, -60000, 60000).astype(np.float16)
    return a16.view(np.uint16), b16.view(np.uint16), c.astype(np.float32)


@pytest.mark.parametrize("kind", ["act", "wide", "tie", "cross", "tiny"])
def test_model_predicts_fresh_instructions(ctx, kind):
    """Operands the model was NOT fitted on (other seeds, other distributions): 4 096 instructions = 1 048 576 outputs per kind, every bit.  (Round 5's first fit passed all 172 032
    probe vectors and failed here on ~1 output in 10^5 — sums crossing a power of two; `cross` now aims at exactly that.)"""
    a, b, c = _operands(np.random.default_rng({"act": 101, "wide": 102, "tie": 103, "tiny": 104, "cross": 105}[kind]), 4096, kind)
    hw = _hw_tiles(ctx, a, b, c); sw = ol.mfma_f16_tiles(a, b, c)
    finite = np.isfinite(hw) & (np.abs(hw) >= 2.0 ** -126) | (hw == 0)                # the header's stated gaps: overflow and f32-subnormal results
    assert finite.mean() > 0.99
    bad = (_bits(hw) != _bits(sw)) & finite
    assert not bad.any(), "%d of %d outputs differ, e.g. hw %r model %r" % (int(bad.sum()), bad.size, hw[bad][:3], sw[bad][:3])


def test_zero_products_and_edge_accumulators_on_the_hardware(ctx):
    """What the CPU tier's test_special_operands claims of the model, asked of the instruction itself: an instruction whose 32 products are all zero returns its accumulator's VALUE
    untouched (+0, an ordinary value, the largest finite f32); a -0 accumulator comes out as +0, also when every product is -0; one zero-sum of non-zero products on a zero
    accumulator gives +0.  (An f32-SUBNORMAL accumulator is outside the model's stated range: reported, not asserted.)"""
    a = np.zeros((1, 16, 32), np.uint16); b = np.zeros((1, 32, 16), np.uint16); a[0, :, ::2] = 0x8000      # products: -0 in the even slots, +0 in the odd ones
    for c in (0.0, 1.5, 3.0e38, -2.5e-30):
        Cc = np.full((1, 16, 16), c, np.float32)
        hw = _hw_tiles(ctx, a, b, Cc)
        assert np.array_equal(_bits(hw), _bits(Cc)) and np.array_equal(_bits(ol.mfma_f16_tiles(a, b, Cc)), _bits(Cc)), c
    mz = np.full((1, 16, 16), -0.0, np.float32)
    for all_minus in (False, True):                           # a zero result is +0 on this hardware, even -0 + (-0) + ... + (-0)
        if all_minus:
            a[:] = 0x8000
        hw = _hw_tiles(ctx, a, b, mz)
        print("accumulator -0, %s zero products: hardware returns %r" % ("all -0" if all_minus else "mixed-sign", float(hw[0, 0, 0])), "(sign bit %d)" % int(_bits(hw)[0, 0, 0] >> 31))
        assert np.array_equal(_bits(hw), _bits(ol.mfma_f16_tiles(a, b, mz))), all_minus
    a[:] = 0; a[0, :, ::2] = 0x8000
    sub = np.full((1, 16, 16), -3.0e-39, np.float32)
    print("subnormal accumulator through an all-zero instruction: hardware returns %r (the model: %r)" % (float(_hw_tiles(ctx, a, b, sub)[0, 0, 0]), float(ol.mfma_f16_tiles(a, b, sub)[0, 0, 0])))
    one = np.float16(1.0).view(np.uint16); a2 = np.zeros((1, 16, 32), np.uint16); b2 = np.zeros((1, 32, 16), np.uint16)
    a2[0, :, 0] = one; a2[0, :, 9] = one | 0x8000; b2[0, 0, :] = one; b2[0, 9, :] = one              # (+1) + (-1) in two different 8-slot groups, C = 0
    z = np.zeros((1, 16, 16), np.float32)
    hw = _hw_tiles(ctx, a2, b2, z)
    assert np.array_equal(_bits(hw), _bits(ol.mfma_f16_tiles(a2, b2, z))) and np.array_equal(_bits(hw), _bits(z))


def _gemm_operands(seed, M, N, K):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, K)) * np.exp2(rng.integers(-1, 4, (M, 1)).astype(np.float64))
    A[:, rng.integers(0, K, max(1, K // 100))] *= 30.0                                 # a few loud channels, as the residual stream has
    W = rng.standard_normal((N, K)) * 0.04
    return A.astype(np.float16).view(np.uint16), W.astype(np.float16).view(np.uint16)


# (M, N, K): one tile and a short contraction / ragged rows / tiny's and small's encoder products (Q, FC1, FC2 of one clip)
ENC_SHAPES = [(128, 256, 128), (200, 384, 256), (1500, 384, 384), (1500, 768, 768), (1500, 3072, 768), (1500, 768, 3072)]


@pytest.mark.parametrize("kernel", ["k_gemm16w", "k_gemm16"])
@pytest.mark.parametrize("shape", ENC_SHAPES, ids=lambda s: "%dx%dx%d" % s)
def test_encoder_gemm_kernels_bit_identical_to_the_oracle(ctx, kernel, shape):
    M, N, K = shape
    A, W = _gemm_operands(M + N + K, M, N, K)
    got = _hw_gemm(ctx, kernel, A, W); want = ol.gemm_f16mfma(A, W, 1)
    bad = _bits(got) != _bits(want)
    assert not bad.any(), "%d of %d outputs differ (max |diff| %g)" % (int(bad.sum()), bad.size, float(np.abs(got - want).max()))


# decode step: (rows = batch lanes, N, K) of small's QKV / O / FC1 / FC2 (four K quarters per output), ragged rows, and the vocabulary product (one chain per output)
DEC_SHAPES = [(64, 2304, 768, 4), (64, 768, 768, 4), (64, 3072, 768, 4), (64, 768, 3072, 4), (5, 384, 384, 4), (37, 768, 1536, 4), (64, 51872, 768, 1), (9, 51872, 384, 1)]


@pytest.mark.parametrize("shape", DEC_SHAPES, ids=lambda s: "%dx%dx%d_split%d" % s)
def test_decode_gemm_kernels_bit_identical_to_the_oracle(ctx, shape):
    M, N, K, split = shape
    A, W = _gemm_operands(M + N + K, M, N, K)
    got = _hw_gemm(ctx, "decode", A, W); want = ol.gemm_f16mfma(A, W, split)
    bad = _bits(got) != _bits(want)
    assert not bad.any(), "%d of %d outputs differ (max |diff| %g)" % (int(bad.sum()), bad.size, float(np.abs(got - want).max()))
    if split == 4 and M >= 16:
        assert (_bits(got) != _bits(ol.gemm_f16mfma(A, W, 1))).any(), "the K split must be visible in the bits, or this case pins nothing about it"
