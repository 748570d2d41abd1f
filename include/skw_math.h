/*
 * skw_math.h — the arithmetic CONTRACT shared by every implementation of the
 * Whisper hot path in this repository (HIP kernels, C++ host code, CPU oracle).
 *
 * Why this header exists
 * ----------------------
 * north_star demands greedy token ids that are bit-exact against the CPU path.
 * The reference delegates all arithmetic to whisper.cpp/ggml
 * (/root/reference/plugins/native/whisper/src/lib.rs:644-646 calls
 * `whisper_state.full`), whose CPU results depend on libm (`expf`, `logf`,
 * `tanhf`) and on SIMD reduction order — i.e. they are not even reproducible
 * between two CPUs.  To make "bit-exact" a checkable property we pin every
 * operation whose result could differ between a CPU and a GPU:
 *
 *   - f32 <-> f16 conversion: IEEE-754 round-to-nearest-even (what ggml's
 *     GGML_FP32_TO_FP16 does through F16C / v_cvt_f16_f32 does on gfx950);
 *   - expf / logf: the polynomial implementations below, built only from
 *     IEEE add/mul/fma/rint and integer bit operations, so that gcc (x86-64)
 *     and hipcc (gfx950) produce identical bits when compiled with
 *     -ffp-contract=off (every fma below is explicit);
 *   - dot products: a k-ascending chain acc = fma(a[k], b[k], acc) — this is
 *     exactly what gfx950's v_mfma_f32_*_f32 computes (MI355X guide,
 *     "FP32-input MFMA": bit-for-bit a k-ordered fmaf chain), and what a plain
 *     scalar C loop computes;
 *     In the one-token decoder graph the K axis is cut into four contiguous
 *     segments, each such a chain from zero, added ((s0+s1)+s2)+s3 (D3');
 *   - long sums that ggml accumulates in `ggml_float` (double): accumulated in
 *     double here too (LayerNorm statistics, softmax denominators), which makes
 *     them insensitive to the reduction order a GPU uses.
 *
 * This header is a public include (the C-ABI library documents its numerics
 * through it).  oracle/ includes it as the specification it restates;
 * the product never includes anything from oracle/.
 */
#ifndef SKW_MATH_H
#define SKW_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SKW_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SKW_HD static inline
#endif

/* ---- bit casts ---------------------------------------------------------- */
SKW_HD uint32_t skw_f32_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
SKW_HD float skw_bits_f32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* ---- f16 conversion (IEEE binary16, round-to-nearest-even) -------------- */
/* Software form; device code may use the hardware cast, which is bit-identical
 * (tests/test_gpu_math.py checks all 2^16 f16 values and a 2^24-point f32 sweep). */
SKW_HD uint16_t skw_f32_to_f16(float f) {
    uint32_t x = skw_f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x0200u : 0u));
    }
    if (ax >= 0x477ff000u) {                       /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) {                        /* < 2^-14: subnormal half or zero */
        if (ax < 0x33000000u) return (uint16_t)sign;   /* < 2^-25 -> 0 (2^-25 ties to even = 0) */
        uint32_t e = ax >> 23;                     /* biased exponent, 102..112 */
        uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
        uint32_t shift = 126u - e;                 /* 14..24: value = m * 2^(e-150); half sub ulp = 2^-24 */
        uint32_t half_m = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
        return (uint16_t)(sign | half_m);
    }
    {
        uint32_t e = (ax >> 23) - 112u;            /* half biased exponent 1..30 */
        uint32_t m = ax & 0x007fffffu;
        uint32_t h = (e << 10) | (m >> 13);
        uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;   /* carry may bump exponent: correct */
        return (uint16_t)(sign | h);
    }
}

SKW_HD float skw_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t out;
    if (e == 0) {
        if (m == 0) out = sign;
        else {                                     /* subnormal: normalise */
            uint32_t sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            out = sign | ((113u - sh) << 23) | (m << 13);
        }
    } else if (e == 31) out = sign | 0x7f800000u | (m << 13);
    else out = sign | ((e + 112u) << 23) | (m << 13);
    return skw_bits_f32(out);
}

/* round a float to the nearest f16-representable float */
SKW_HD float skw_round_f16(float f) { return skw_f16_to_f32(skw_f32_to_f16(f)); }

/* ---- expf --------------------------------------------------------------- */
/* Cody-Waite reduction + degree-5 minimax polynomial (Cephes coefficients).
 * Domain notes: returns 0 for x < -86 (results would leave the normal range;
 * every caller passes x <= 0 and treats such terms as 0), +inf for x > 88.7. */
SKW_HD float skw_expf(float x) {
    if (!(x >= -86.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.72283f) return skw_bits_f32(0x7f800000u);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    if (ni > 127) { y = y * 2.0f; ni -= 1; }
    return y * skw_bits_f32((uint32_t)(ni + 127) << 23);
}

/* ---- logf (x > 0, normal) ----------------------------------------------- */
/* Cephes logf: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); log(m) by polynomial. */
SKW_HD float skw_logf(float x) {
    uint32_t ux = skw_f32_bits(x);
    int e = (int)(ux >> 23) - 126;                         /* x = m * 2^e, m in [0.5,1) */
    float m = skw_bits_f32((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

/* ---- GELU as ggml evaluates it on CPU ------------------------------------ */
/* ggml's ggml_vec_gelu_f32 (GGML_GELU_FP16 build, the default):
 *   x <= -10 -> 0;  x >= 10 -> x;  else f16 table lookup indexed by f16(x),
 *   table[i] = f16( 0.5*x*(1+tanhf(0.79788456*x*(1+0.044715*x*x))) ), x = f32(f16 bits i).
 * The table is built once on the host (skw_gelu_table_entry, libm tanhf) by
 * whoever needs it; lookups are then exact everywhere. */
#define SKW_GELU_COEF_A 0.044715f
#define SKW_SQRT_2_OVER_PI 0.79788456080286535587989211986876f

#include <math.h>
#if defined(__HIPCC__)
#define SKW_H __host__ inline
#else
#define SKW_H static inline
#endif
SKW_H uint16_t skw_gelu_table_entry(uint16_t i) {
    float x = skw_f16_to_f32(i);
    float g = 0.5f * x * (1.0f + tanhf(SKW_SQRT_2_OVER_PI * x * (1.0f + SKW_GELU_COEF_A * x * x)));
    return skw_f32_to_f16(g);
}

/* gelu through the table: returns an f32 (f16-valued unless x >= 10) */
SKW_HD float skw_gelu_lookup(float x, const uint16_t* table) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    return skw_f16_to_f32(table[skw_f32_to_f16(x)]);
}

#endif /* SKW_MATH_H */
