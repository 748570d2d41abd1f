"""k_gemm16w — the encoder's GEMM since round 4: weights straight from fragment-order images into MFMA operands, tokens through an LDS ring by LDS-DMA, both
vector-memory queues counted by hand (`s_waitcnt vmcnt(8)` / `vmcnt(4)`) — held BIT FOR BIT to k_gemm16 (both operands through LDS, compiler-counted waits) on seeded operands,
for every epilogue, ragged and whole tile counts, short and long contractions, and every feature-split tile walk.  This is the regression cover of the round-4 memory fault
(look-ahead weight loads still in flight when their registers were reused: `skw_kernels_f16.hip`, the final wait of the K loop): a later edit that miscounts a queue changes bits
here before it faults anywhere.  One product per case through `skw_debug_gemm16_compare` (streamkit_amd/csrc/skw_engine.hip), which launches both kernels and compares every byte."""
import ctypes as C

import numpy as np
import pytest

from streamkit_amd import synth

pytestmark = pytest.mark.gpu

EPI = {"F32": 0, "F16_KPERM": 1, "GELU_F16_KPERM": 2, "CONV2": 3, "HEADS_F16": 4, "VT_F16": 5, "F16_PLAIN": 6, "GELU_F16_KPERM_ROWPAD": 7}
N_CTX, TPAD = 300, 320      # rows per clip of the per-clip layouts in these cases (a tile is shorter than a clip: the kernels' own precondition)


@pytest.fixture(scope="module")
def ctx(tiny_model_path):
    from streamkit_amd import engine
    m = engine.Model(tiny_model_path); c = engine.Context(m, max_batch=1, max_samples=16000)
    c.set_precision("f16_mfma")
    L = engine.lib()
    L.skw_debug_gemm16_compare.restype = C.c_long
    L.skw_debug_gemm16_compare.argtypes = [C.c_void_p] + [C.c_int] * 9
    yield c, L
    c.close(); m.close()


def _cmp(ctx, M, N, K, epi, frag=0, ngroups=0, res=1):
    c, L = ctx
    r = L.skw_debug_gemm16_compare(c.h, M, N, K, EPI[epi], frag, N_CTX, TPAD, ngroups, res)
    assert r != -1, c.last_error()
    return r


# (M, N, K): one 128-row tile / ragged rows / many tiles with a ragged last one; narrow, odd-tile-count and > 4096 features (the bias table's fallback); one, few and many K steps
ROW_MAJOR_SHAPES = [(128, 384, 64), (200, 1280, 256), (4500, 5120, 64), (1500, 768, 1280), (640, 2304, 768), (4500, 384, 3072)]


@pytest.mark.parametrize("epi", ["F32", "F16_KPERM", "GELU_F16_KPERM", "F16_PLAIN"])
@pytest.mark.parametrize("shape", ROW_MAJOR_SHAPES, ids=lambda s: "%dx%dx%d" % s)
def test_row_major_epilogues_bit_identical(ctx, epi, shape):
    M, N, K = shape
    assert _cmp(ctx, M, N, K, epi) == 0
    if epi == "F32":
        assert _cmp(ctx, M, N, K, epi, res=0) == 0


@pytest.mark.parametrize("epi", ["HEADS_F16", "VT_F16", "CONV2", "GELU_F16_KPERM_ROWPAD"])
@pytest.mark.parametrize("clips, N, K", [(1, 384, 256), (3, 768, 768), (15, 1280, 64), (5, 768, 3072)])
def test_per_clip_epilogues_bit_identical(ctx, epi, clips, N, K):
    """M = whole clips of N_CTX rows that are NOT multiples of the tile height (300 rows: a tile straddles clips; V^T: pad keys 300 .. 320 stay zero)."""
    assert _cmp(ctx, clips * N_CTX, N, K, epi) == 0


@pytest.mark.parametrize("epi", ["F16_PLAIN", "VT_F16"])
@pytest.mark.parametrize("clips, N, K", [(2, 384, 384), (7, 768, 768)])
def test_fragment_order_cross_kv_images_bit_identical(ctx, epi, clips, N, K):
    """the f16_mfma precision's cross K / V^T: the same products written as fragment-order images (skw_kfrag_off / skw_vtfrag_off)"""
    assert _cmp(ctx, clips * N_CTX, N, K, epi, frag=1) == 0


@pytest.mark.parametrize("ngroups", [1, 2, 4, 8])
@pytest.mark.parametrize("epi, shape", [("GELU_F16_KPERM", (4500, 3072, 768)), ("GELU_F16_KPERM", (1500, 4096, 1024)), ("F32", (4500, 2048, 512)), ("F16_KPERM", (3000, 5120, 1280))],
                         ids=lambda v: v if isinstance(v, str) else "%dx%dx%d" % v)
def test_feature_split_tile_walks_bit_identical(ctx, ngroups, epi, shape):
    """GEMM16W_NGROUPS = 1 / 2 / 4 / 8: the XCDs split the features into n-tile groups when the weight does not fit an XCD's L2 (FC1).  A walk changes which
    workgroup computes which tile, never a tile's arithmetic: every forced walk (the launcher falls back to the contiguous walk where the tile count does not divide) equals
    k_gemm16."""
    M, N, K = shape
    assert _cmp(ctx, M, N, K, epi, ngroups=ngroups) == 0


def test_geometries_outside_k_gemm16w_are_reported_not_compared(ctx):
    """M < 128 is k_gemm16's: the compare entry says so (-2) instead of comparing a kernel with itself"""
    assert _cmp(ctx, 64, 384, 256, "F32") == -2


def test_encoder_and_cross_kv_identical_with_and_without_k_gemm16w(eng, tiny_model_path):
    """End to end: encoder output and cross K / V^T taps of a 30 s clip, and a ragged batch's f16_mfma transcripts (tokens and log-probs), with the switch GEMM16W = 0
    (every big GEMM through k_gemm16) and with the default — bit-identical."""
    pcm = synth.clip(3, 16000 * 30)
    pcms = [pcm, synth.clip(5, 16000 * 9), synth.clip(11, 16000 * 41)]
    out = {}
    for on in (1, 0):
        with eng.switch("GEMM16W", on):
            m = eng.Model(tiny_model_path); c = eng.Context(m, max_batch=4, max_samples=16000 * 45); c.set_precision("f16_mfma")
            enc, ck, cv = c.encode(pcm)
            res = c.full_batch(pcms)
            out[on] = (np.asarray(enc).copy(), np.asarray(ck).copy(), np.asarray(cv).copy(), [[(t[0], t[3]) for t in r["tokens"]] for r in res])
            c.close(); m.close()
    for a, b, name in zip(out[1][:3], out[0][:3], ("enc_out", "cross_k", "cross_v")):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    assert out[1][3] == out[0][3]
