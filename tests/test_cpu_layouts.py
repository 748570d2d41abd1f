"""CPU tier: the fragment-order cross K / V^T layouts (streamkit_amd/csrc/skw_kernels.h: skw_kfrag_off, skw_vtfrag_off) as the host sees them — no GPU, no compute.
f16_mfma's decode-step cross attention (k_dec_cross_attn16) assumes: every 16-byte chunk of the row layouts has exactly one home in the image; a wave-load's 64 chunks are one
contiguous KiB; lane (i, g) of a K tile's load holds key 16 T + 4 (i & 3) + (i >> 2), d = 32 kk + 8 g; lane (i, g) of a V^T tile's load holds channel 16 ct + i, positions 8 g .. of the block;
and the score accumulators' key order (lane group g, element e = 4 j + r <-> key 4 e + g of the 32-key block) is the order V^T's positions are stored in (skw_kperm)."""
import ctypes as C

import numpy as np
import pytest

from streamkit_amd import engine


@pytest.fixture(scope="module")
def L():
    lib = engine.lib()
    lib.skw_layout_kfrag_off.restype = C.c_long; lib.skw_layout_kfrag_off.argtypes = [C.c_int] * 5
    lib.skw_layout_vtfrag_off.restype = C.c_long; lib.skw_layout_vtfrag_off.argtypes = [C.c_int] * 5
    lib.skw_layout_kperm.restype = C.c_int; lib.skw_layout_kperm.argtypes = [C.c_int]
    return lib


@pytest.mark.parametrize("H,n_ctx", [(12, 1500), (6, 1500), (8, 96), (20, 1500)])
def test_fragment_layouts_are_chunk_permutations(L, H, n_ctx):
    Tpad = (n_ctx + 31) & ~31
    d = 64 * H
    for slot in (0, 3):
        seen_k = set(); seen_v = set()
        for key in range(0, Tpad, 1 if Tpad <= 128 else 7):
            for feat in range(0, d, 8):
                o = L.skw_layout_kfrag_off(slot, H, Tpad, key, feat)
                assert o % 8 == 0 and slot * Tpad * d <= o < (slot + 1) * Tpad * d
                assert o not in seen_k; seen_k.add(o)
        for feat in range(0, d, 1 if d <= 512 else 5):
            for pos in range(0, Tpad, 8):
                o = L.skw_layout_vtfrag_off(slot, H, Tpad, feat, pos)
                assert o % 8 == 0 and slot * Tpad * d <= o < (slot + 1) * Tpad * d
                assert o not in seen_v; seen_v.add(o)


def test_a_wave_load_is_one_contiguous_kib_in_mfma_operand_order(L):
    H, Tpad = 12, 1504
    for h in (0, 5, 11):
        for T in (0, 1, 93):            # K: 16-key tile T, d half kk -> 1 KiB at ((h * Tpad/16 + T) * 2 + kk) KiB of the slot
            for kk in (0, 1):
                base = ((h * (Tpad // 16) + T) * 2 + kk) * 512
                for lane in range(64):
                    i, g = lane & 15, lane >> 4
                    key = 16 * T + 4 * (i & 3) + (i >> 2)
                    assert L.skw_layout_kfrag_off(0, H, Tpad, key, 64 * h + 32 * kk + 8 * g) == base + lane * 8
        for kb in (0, 46):              # V^T: 32-key block kb, channel tile ct -> 1 KiB at ((h * Tpad/32 + kb) * 4 + ct) KiB
            for ct in range(4):
                base = ((h * (Tpad // 32) + kb) * 4 + ct) * 512
                for lane in range(64):
                    i, g = lane & 15, lane >> 4
                    assert L.skw_layout_vtfrag_off(0, H, Tpad, 64 * h + 16 * ct + i, 32 * kb + 8 * g) == base + lane * 8


def test_score_accumulator_order_is_the_stored_key_order(L):
    # score tile j of a 32-key block: MFMA output row 4 g + r of lane group g is A row i = 4 g + r, i.e. key 16 j + 4 (i & 3) + (i >> 2) = 16 j + 4 r + g.
    # As the P.V operand, lane group g element e = 4 j + r multiplies the V^T value at memory position 8 g + e of the block, which holds logical key kperm^-1(8 g + e).
    inv = {L.skw_layout_kperm(k): k for k in range(32)}
    assert sorted(inv) == list(range(32))
    for g in range(4):
        for j in range(2):
            for r in range(4):
                i = 4 * g + r
                key_from_scores = 16 * j + 4 * (i & 3) + (i >> 2)
                assert key_from_scores == inv[8 * g + 4 * j + r]


def test_activation_image_offsets(L):
    # skw_afrag_off: the fc1 -> fc2 hand-over and the attention outputs (SkwGemmArgs::c_frag / a_frag): a row tile's 32-k block is one KiB, lane r16 + 16 g holds row 16 t + r16, positions 8 g .. 8 g + 7
    L.skw_layout_afrag_off.restype = C.c_long; L.skw_layout_afrag_off.argtypes = [C.c_int] * 3
    K = 3072
    seen = set()
    for m in range(0, 48):
        for p in range(0, K, 4):
            o = L.skw_layout_afrag_off(m, p, K)
            assert o % 4 == 0 and 0 <= o < 48 * K and o not in seen
            seen.add(o)
    for t in (0, 2):
        for kb in (0, 5, 95):
            base = (t * (K // 32) + kb) * 512
            for lane in range(64):
                r16, g = lane & 15, lane >> 4
                assert L.skw_layout_afrag_off(16 * t + r16, 32 * kb + 8 * g, K) == base + lane * 8
                assert L.skw_layout_afrag_off(16 * t + r16, 32 * kb + 8 * g + 4, K) == base + lane * 8 + 4
