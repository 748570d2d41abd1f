// skw_dev_common.h — device helpers shared by the exact (skw_kernels.hip) and the f16-MFMA (skw_kernels_f16.hip) kernels:
// conversions with pinned rounding, ggml's GELU table lookup, and the GEMM epilogues (identical in both precisions: only the
// contraction differs).
#pragma once
#include "skw_kernels.h"
#include "../../include/skw_math.h"
#include "../../include/skw_ggml_quant.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
// f32 -> f16 of an ALREADY ROUNDED f32 value.  The empty asm makes the operand opaque: without it hipcc folds
// `(half)(a * b)` / `(half)(a + b)` into v_fma_mixlo_f16, which rounds the exact product once (to f16) instead of
// twice (f32, then f16) and so differs from ggml's f32 -> f16 conversion of an f32 result on ties-after-rounding.
__device__ __forceinline__ half_t f2h(float f) { asm("" : "+v"(f)); return (half_t)f; }   // v_cvt_f16_f32, RNE
__device__ __forceinline__ float gelu_dev(float x, const uint16_t* tab) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    half_t h = f2h(x); uint16_t bits = __builtin_bit_cast(uint16_t, h);
    uint16_t o = tab[bits];
    return h2f(__builtin_bit_cast(half_t, o));
}

union H8 { uint4 u; half_t h[8]; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union H8v { u32x4 v; half_t h[8]; };

// ------------------------------------------------------------------ LayerNorm of R rows by one wave (K3: ggml_norm + mul + add)
// Lane l owns elements l, l + 64, ... of each row (d <= 1536).  Statistics as ggml_norm: f64 sums over the row, mean and variance
// rounded to f32, then scale, gain, bias; the f16 image is written in kperm order.  k_layernorm, the embedding kernel and the
// LayerNorm tail of the decode GEMMs all call this, so every one of them produces the same bits for the same row.
// Wave-wide reductions without the LDS crossbar.  `__shfl_xor` compiles to ds_bpermute_b32 — an LDS round trip per level (two for a
// double), six dependent levels per reduction: ~0.3 us, and a decode-step LayerNorm (4.9 us, launch included) does two.  Here the
// four levels inside a row of 16 lanes are DPP moves (quad_perm [1,0,3,2] and [2,3,0,1], then row_half_mirror and row_mirror: after
// the quad stages every lane of a quad holds the same value, so "the lane mirrored across the half row" is "the other quad"), and the
// four row totals are read with v_readlane and combined as the xor-16 / xor-32 levels would: (r0 + r1) + (r2 + r3) — the same tree in
// every lane, addition being commutative.  All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ int skw_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float skw_dpp_f32(float v) { return __int_as_float(skw_dpp_i32<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double skw_dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = skw_dpp_i32<CTRL>((int)(b & 0xffffffffll)), hi = skw_dpp_i32<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double skw_readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double skw_wave_sum_f64(double v) {
    v += skw_dpp_f64<0xB1>(v); v += skw_dpp_f64<0x4E>(v); v += skw_dpp_f64<0x141>(v); v += skw_dpp_f64<0x140>(v);
    const double r0 = skw_readlane_f64(v, 0), r1 = skw_readlane_f64(v, 16), r2 = skw_readlane_f64(v, 32), r3 = skw_readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}
// The xor-16 and xor-32 levels for values that differ per lane of a row (one query per lane): gfx950's v_permlane16_swap / v_permlane32_swap
// exchange odd rows of one register with even rows of another (upper and lower half waves for the 32 form); fed the same value twice they
// leave (own row pair's even row, own row pair's odd row) in the two results, whose max / sum is what `op(v, __shfl_xor(v, 16))` gives.
__device__ __forceinline__ float skw_rows_max_f32(float v) {      // max over the four lanes l, l ^ 16, l ^ 32, l ^ 48
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float skw_rows_sum_f32(float v) {      // (v[l] + v[l ^ 16]) + (v[l ^ 32] + v[l ^ 48]), the order of the two shuffle levels
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float skw_wave_max_f32(float v) {
    v = fmaxf(v, skw_dpp_f32<0xB1>(v)); v = fmaxf(v, skw_dpp_f32<0x4E>(v)); v = fmaxf(v, skw_dpp_f32<0x141>(v)); v = fmaxf(v, skw_dpp_f32<0x140>(v));
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// NC slots of 64 elements per row: 12 covers d <= 768 with half the instructions (same operations on the live elements); FULL: d == 64 NC, no tail predicates
template <int R, int NC = 24, bool FULL = false>
__device__ __forceinline__ void skw_ln_rows(float (&v)[R][NC], const float (&wv)[NC], const float (&bv)[NC], int d, int lane, const bool (&live)[R],
    half_t* const (&out16)[R], float* const (&out32)[R]) {
    double sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sum[r] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) sum[r] += (double)v[r][c];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = skw_wave_sum_f64(sum[r]);
    double sum2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float mean = (float)(sum[r] / (double)d);
        sum2[r] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const int i = lane + 64 * c; if (FULL || i < d) { const float t = v[r][c] - mean; v[r][c] = t; sum2[r] += (double)(t * t); } }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) sum2[r] = skw_wave_sum_f64(sum2[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float variance = (float)(sum2[r] / (double)d);
        const float scale = 1.0f / sqrtf(variance + 1e-5f);
        if (!live[r]) continue;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + 64 * c;
            if (FULL || i < d) {
                float t = v[r][c] * scale; t = t * wv[c]; t = t + bv[c];
                v[r][c] = t;                                    // (left in place for callers that go on with the normalised row: k_layernorm_q8)
                if (out16[r]) out16[r][skw_kperm(i)] = f2h(t);
                if (out32[r]) out32[r][i] = t;
            }
        }
    }
}

// One 32-block held one value per lane by 32 consecutive lanes (lanes 0-31 or 32-63 of a wave): quantize_row_q8_* with the maximum and
// the sum taken across the half wave (max and integer sum are order-free, so the bits are k_q8_quantize's).  Every lane returns q; d and s
// are valid in all 32 lanes.
__device__ __forceinline__ int q8_half_wave_block(float x, float* d_out, float* s_out) {
    float amax = x < 0.0f ? -x : x;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float d = amax / 127.0f, idv = d != 0.0f ? 1.0f / d : 0.0f;
    const int q = (int)skw_roundf(x * idv);
    int sum = q;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    *d_out = skw_round_f16(d); *s_out = skw_round_f16((float)sum * d);
    return q;
}

// ------------------------------------------------------------------ epilogues
template <int EPI>
__device__ __forceinline__ void epi_store(const SkwGemmArgs& a, int m, int n, float v) {
    if (EPI == EPI_F32) {
        if (a.bias) v = v + a.bias[n];
        if (a.res) v = v + a.res[(long)m * a.ldres + n];
        ((float*)a.C)[(long)m * a.ldc + n] = v;
    } else if (EPI == EPI_F16_KPERM) {
        if (a.bias) v = v + a.bias[n];
        if (a.has_scale) v = v * a.scale;
        ((half_t*)a.C)[(long)m * a.ldc + skw_kperm(n)] = f2h(v);
    } else if (EPI == EPI_GELU_F16_KPERM) {
        if (a.bias) v = v + a.bias[n];
        ((half_t*)a.C)[(long)m * a.ldc + skw_kperm(n)] = f2h(gelu_dev(v, a.gelu_tab));
    } else if (EPI == EPI_GELU_F16_KPERM_ROWPAD) {
        if (a.bias) v = v + a.bias[n];
        long row = (long)(m / a.n_ctx) * (a.n_ctx + 2) + (m % a.n_ctx) + 1;
        ((half_t*)a.C)[row * a.ldc + skw_kperm(n)] = f2h(gelu_dev(v, a.gelu_tab));
    } else if (EPI == EPI_CONV2) {
        if (a.bias) v = v + a.bias[n];
        float g = gelu_dev(v, a.gelu_tab);
        ((float*)a.C)[(long)m * a.ldc + n] = a.pe[(long)(m % a.n_ctx) * a.N + n] + g;
    } else if (EPI == EPI_GELU_F32) {
        if (a.bias) v = v + a.bias[n];
        ((float*)a.C)[(long)m * a.ldc + n] = gelu_dev(v, a.gelu_tab);
    } else if (EPI == EPI_HEADS_F16) {
        if (a.bias) v = v + a.bias[n];
        if (a.has_scale) v = v * a.scale;
        int b = m / a.n_ctx, i = m % a.n_ctx, h = n >> 6, d = n & 63;
        ((half_t*)a.C)[((long)(b * a.H + h) * a.Tpad + i) * 64 + skw_kperm(d)] = f2h(v);
    } else if (EPI == EPI_VT_F16) {
        if (a.bias) v = v + a.bias[m];
        int b = n / a.n_ctx, key = n % a.n_ctx, h = m >> 6, c = m & 63;
        ((half_t*)a.C)[((long)(b * a.H + h) * 64 + c) * a.Tpad + skw_kperm(key)] = f2h(v);
    } else if (EPI == EPI_DEC_QKV) {
        // fused decoder Q|K|V projection: n in [0,d) -> q (plain, +bias, *scale); [d,2d) -> K cache (*scale); [2d,3d) -> V cache (+bias)
        const int d = a.n_ctx;
        if (a.bias) v = v + a.bias[n];
        if (n < 2 * d) v = v * a.scale;
        if (n < d) ((half_t*)a.C)[(long)m * a.ldc + n] = f2h(v);
        else {
            const long po = a.pos_ptr ? (long)a.pos_ptr[(long)m * a.pos_stride] * d : 0;
            if (n < 2 * d) ((half_t*)a.C2)[(long)m * a.ldc2 + po + (n - d)] = f2h(v);
            else ((half_t*)a.C3)[(long)m * a.ldc2 + po + (n - 2 * d)] = f2h(v);
        }
    } else if (EPI == EPI_F16_PLAIN) {
        if (a.bias) v = v + a.bias[n];
        if (a.has_scale) v = v * a.scale;
        ((half_t*)a.C)[(long)m * a.ldc + n] = f2h(v);
    }
}

