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
 * DEVIATION D4: ggml multiplies quantised weights by activations it first quantises to q8 blocks (integer dot
 * products, per-block scales).  Here the weights are dequantised once (f32: q*d exact, +m one rounding), rounded to f16 (RNE) and
 * run through the f16-weight path — the transcripts of a quantised file are therefore those of its dequantised f16 twin, within
 * the quantisation noise of ggml's own result, not bit-identical to it.  Engine and oracle share this header, so they agree bitwise.
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

/* n values (a multiple of 32) from blocks to f16 bit patterns (RNE of the f32 value) */
static inline void skw_ggml_dequant_to_f16(int type, const uint8_t* blocks, size_t n, uint16_t* out) {
    const size_t bb = skw_ggml_block_bytes(type); float y[32];
    for (size_t i = 0; i < n / 32; ++i) { skw_ggml_dequant_block(type, blocks + i * bb, y); for (int j = 0; j < 32; ++j) out[i * 32 + j] = skw_f32_to_f16(y[j]); }
}
#endif
