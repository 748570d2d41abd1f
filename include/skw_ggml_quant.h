/* skw_ggml_quant.h — GGML block-quantised tensor formats, decoded at load time.
 *
 * whisper.cpp's model files may carry their 2-D weight matrices as q4_0 / q4_1 / q5_0 / q5_1 / q8_0 blocks of 32 values
 * (the reference's default model is ggml-base.en-q5_1.bin: /root/reference/plugins/native/whisper/src/lib.rs:115, :247).
 * Block layouts follow ggml-common.h (third-party, not vendored in the reference; whisper-rs 0.15 / whisper.cpp 1.7.x):
 *
 *   q4_0  { f16 d;           u8 qs[16]; }  x = (nibble - 8)  * d
 *   q4_1  { f16 d; f16 m;    u8 qs[16]; }  x =  nibble       * d + m
 *   q5_0  { f16 d; u8 qh[4]; u8 qs[16]; }  x = (5-bit  - 16) * d
 *   q5_1  { f16 d; f16 m; u8 qh[4]; u8 qs[16]; }  x = 5-bit * d + m
 *   q8_0  { f16 d;           i8 qs[32]; }  x =  qs           * d
 *   element j < 16 takes the low nibble of qs[j] (+ bit j of qh), element j + 16 the high nibble (+ bit j + 16 of qh).
 *
 * Two arithmetics for a quantised file:
 *
 *  (a) ggml's own — the default in the exact precision.  ggml_compute_forward_mul_mat converts each f32 activation row to the weight
 *      type's vec_dot_type (q8_0 for q4_0 / q5_0 / q8_0, q8_1 for q4_1 / q5_1: quantize_row_q8_* below) and takes per-block integer dot
 *      products with f32 scales.  Restated from ggml-quants.c (quantize_row_q8_0_ref / q8_1_ref, ggml_vec_dot_*_generic; third-party,
 *      NOT in /root/reference: recalled, flagged in DESIGN.md):
 *          q8 block of 32:  amax = max |x|;  d = amax / 127;  id = d ? 1/d : 0;  qs[j] = roundf(x[j] * id);  stored d = f16(d);
 *                           q8_1 also stores s = f16(d * sum(qs))   (the UNROUNDED d; whisper.cpp >= 1.5.5: d and s are f16)
 *          per block b (ascending), sumi = sum_j qw[j] * qs[j] in int32, then
 *              q4_0:          sumf += ((float)sumi * dw) * dy                     qw = nibble - 8
 *              q5_0, q8_0:    sumf += (dw * dy) * (float)sumi                     qw = 5-bit - 16 / int8
 *              q4_1, q5_1:    sumf += ((dw * dy) * (float)sumi + mw * sy)         qw = nibble / 5-bit, unsigned
 *      every product and sum rounded to f32 separately (no contraction).  What cannot be restated is the SIMD build's order of the
 *      block sums (AVX2 keeps eight partial sums): like D3 this is a chain of our choosing, block-ascending — in the one-token decoder
 *      graph cut into four contiguous runs of blocks whose partial sums are added in ascending order (D3', as for f16 weights).  get_rows on the
 *      quantised token embedding dequantises (q * d, + m) to f32 without rounding to f16.
 *
 *  (b) the dequantised f16 twin — what the f16_mfma precision runs, and what SKW_QUANT_TWIN=1 / quant_mode 0 selects everywhere:
 *      weights dequantised once (f32: q*d exact, +m one rounding), rounded to f16 (RNE) and run through the f16-weight path.
 *      Transcripts are those of the file's f16 twin, within the quantisation noise of (a), not identical to it.
 *  Engine and oracle share this header, so they agree bitwise in either arithmetic.
 */
#ifndef SKW_GGML_QUANT_H
#define SKW_GGML_QUANT_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "skw_math.h"

enum { SKW_GGML_F32 = 0, SKW_GGML_F16 = 1, SKW_GGML_Q4_0 = 2, SKW_GGML_Q4_1 = 3, SKW_GGML_Q5_0 = 6, SKW_GGML_Q5_1 = 7, SKW_GGML_Q8_0 = 8 };

/* bytes per 32-value block; 0 = not a supported quantised type */
static inline size_t skw_ggml_block_bytes(int type) {
    switch (type) { case SKW_GGML_Q4_0: return 18; case SKW_GGML_Q4_1: return 20; case SKW_GGML_Q5_0: return 22; case SKW_GGML_Q5_1: return 24; case SKW_GGML_Q8_0: return 34; default: return 0; }
}

static inline void skw_ggml_dequant_block(int type, const uint8_t* b, float* y /* [32] */) {
    uint16_t dh, mh = 0; uint32_t qh = 0; const uint8_t* qs;
    memcpy(&dh, b, 2); b += 2;
    if (type == SKW_GGML_Q4_1 || type == SKW_GGML_Q5_1) { memcpy(&mh, b, 2); b += 2; }
    if (type == SKW_GGML_Q5_0 || type == SKW_GGML_Q5_1) { memcpy(&qh, b, 4); b += 4; }
    qs = b;
    const float d = skw_f16_to_f32(dh), m = skw_f16_to_f32(mh);
    if (type == SKW_GGML_Q8_0) { for (int j = 0; j < 32; ++j) y[j] = (float)(int8_t)qs[j] * d; return; }
    for (int j = 0; j < 16; ++j) {
        int x0 = qs[j] & 0x0F, x1 = qs[j] >> 4;
        if (type == SKW_GGML_Q5_0 || type == SKW_GGML_Q5_1) { x0 |= (int)((qh >> j) & 1u) << 4; x1 |= (int)((qh >> (j + 16)) & 1u) << 4; }
        switch (type) {
            case SKW_GGML_Q4_0: y[j] = (float)(x0 - 8) * d;  y[j + 16] = (float)(x1 - 8) * d;  break;
            case SKW_GGML_Q5_0: y[j] = (float)(x0 - 16) * d; y[j + 16] = (float)(x1 - 16) * d; break;
            default: { float p0 = (float)x0 * d, p1 = (float)x1 * d; y[j] = p0 + m; y[j + 16] = p1 + m; }   /* q4_1, q5_1 */
        }
    }
}

/* a block in the common form the integer dot uses: 32 signed bytes qw, scale d, offset m (0 for the symmetric types) */
static inline void skw_ggml_unpack_block(int type, const uint8_t* b, int8_t* qw /* [32] */, float* d, float* m) {
    uint16_t dh, mh = 0; uint32_t qh = 0; const uint8_t* qs;
    memcpy(&dh, b, 2); b += 2;
    if (type == SKW_GGML_Q4_1 || type == SKW_GGML_Q5_1) { memcpy(&mh, b, 2); b += 2; }
    if (type == SKW_GGML_Q5_0 || type == SKW_GGML_Q5_1) { memcpy(&qh, b, 4); b += 4; }
    qs = b; *d = skw_f16_to_f32(dh); *m = skw_f16_to_f32(mh);
    if (type == SKW_GGML_Q8_0) { for (int j = 0; j < 32; ++j) qw[j] = (int8_t)qs[j]; return; }
    for (int j = 0; j < 16; ++j) {
        int x0 = qs[j] & 0x0F, x1 = qs[j] >> 4;
        if (type == SKW_GGML_Q5_0 || type == SKW_GGML_Q5_1) { x0 |= (int)((qh >> j) & 1u) << 4; x1 |= (int)((qh >> (j + 16)) & 1u) << 4; }
        if (type == SKW_GGML_Q4_0) { x0 -= 8; x1 -= 8; } else if (type == SKW_GGML_Q5_0) { x0 -= 16; x1 -= 16; }
        qw[j] = (int8_t)x0; qw[j + 16] = (int8_t)x1;
    }
}
/* which of the three per-block forms above a weight type uses: 1 = q4_0, 2 = q5_0 / q8_0, 3 = q4_1 / q5_1 (q8_1 activations) */
SKW_HD int skw_ggml_dot_form(int type) { return type == SKW_GGML_Q4_0 ? 1 : (type == SKW_GGML_Q4_1 || type == SKW_GGML_Q5_1) ? 3 : (type == SKW_GGML_Q5_0 || type == SKW_GGML_Q8_0) ? 2 : 0; }
/* roundf (half away from zero) without libm: exact for |x| < 2^23, and q8 arguments are within [-127, 127] */
SKW_HD float skw_roundf(float x) { float t = (float)(int)x; float r = x - t; if (r >= 0.5f) t += 1.0f; else if (r <= -0.5f) t -= 1.0f; return t; }
/* quantize_row_q8_0 / q8_1 of one 32-value block: qs, d (f16-rounded, as stored), s (f16-rounded d_unrounded * sum; q8_1 only) */
SKW_HD void skw_ggml_quantize_q8_block(const float* x, int8_t* qs, float* d_out, float* s_out) {
    float amax = 0.0f;
    for (int j = 0; j < 32; ++j) { const float v = x[j] < 0.0f ? -x[j] : x[j]; if (v > amax) amax = v; }
    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    int sum = 0;
    for (int j = 0; j < 32; ++j) { const float v = x[j] * id; const int q = (int)skw_roundf(v); qs[j] = (int8_t)q; sum += q; }
    *d_out = skw_round_f16(d); *s_out = skw_round_f16((float)sum * d);
}
/* one block's contribution, added to the running sum */
SKW_HD float skw_ggml_block_dot(int form, float sumf, int sumi, float dw, float mw, float dy, float sy) {
    if (form == 1) { float t = (float)sumi * dw; t = t * dy; return sumf + t; }
    float dd = dw * dy; float t = dd * (float)sumi;
    if (form == 3) { float u = mw * sy; t = t + u; }
    return sumf + t;
}

/* n values (a multiple of 32) from blocks to f16 bit patterns (RNE of the f32 value) */
static inline void skw_ggml_dequant_to_f16(int type, const uint8_t* blocks, size_t n, uint16_t* out) {
    const size_t bb = skw_ggml_block_bytes(type); float y[32];
    for (size_t i = 0; i < n / 32; ++i) { skw_ggml_dequant_block(type, blocks + i * bb, y); for (int j = 0; j < 32; ++j) out[i * 32 + j] = skw_f32_to_f16(y[j]); }
}
#endif
