/*
 * skw_kokoro_oracle.c — CPU restatement of the reduced Kokoro-shaped synthesiser (TEST INFRASTRUCTURE; nothing under streamkit_amd/ links this).
 *
 * PARITY UNPINNED: what the reference runs for this node is Kokoro-82M's ONNX graph inside onnxruntime, reached through sherpa-onnx
 * (/root/reference/plugins/native/kokoro/src/ffi.rs:119-137; call site kokoro_node.rs:581-588).  Neither the graph, the weights nor the
 * runtime are in /root/reference or offline, and no reference test holds an audio vector.  This file restates, in plain C, the network
 * streamkit_amd/csrc/skw_tts.hip evaluates (specified in DESIGN.md section 7) so that the HIP kernels have a checker:
 *   embedding -> n_te x [conv1d k5, LayerNorm, LeakyReLU 0.2] -> AdaLN(prosody style) -> durations = max(1, rint(sum_k sigmoid(.) * scale))
 *   -> length regulation -> F0 = 60 + 340 sigmoid(.), energy -> conv1d k3 + AdaIN(acoustic style) + n_dec residual AdaIN blocks
 *   -> ConvTranspose1d(k = stride = 120) + harmonic source -> snake ResBlock -> LeakyReLU 0.01 -> conv_post k7 -> exp / sin -> iSTFT(20, 5, Hann)
 * Every contraction accumulates in f64 in ascending (tap, input channel) order and rounds once (the rule of include/skw_math.h for ggml_norm);
 * sigmoid and exp use skw_expf, which is bit-identical on host and device.  Weights arrive as plain arrays in their ONNX layouts
 * (tests/kokoro_lib.py reads the file with tests/onnx_mini.py — a reader of its own); the product has its own reader and its own tokeniser.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/skw_math.h"

#define U_ 120
#define NFFT_ 20
#define HOP_ 5
#define BINS_ 11
#define H_ 8
#define STYLE_ 128
#define SUBRATE_ 4800.0

typedef struct { int T, d, n_te, K, C, n_dec, G; float scale; int max_frames; } skwo_tts_dims;
/* weights in this order: emb; per te block: w [d][d][5], b, gamma, beta; pfc_w [2d][128], pfc_b; dur_w [K][d], dur_b; f0_w [d], f0_style [128], f0_b [1];
 * n_w [d], n_b [1]; enc_w [C][d+2][3], enc_b, encfc_w [2C][128], encfc_b; per dec block: w [C][C][3], b, fc_w [2C][128], fc_b;
 * ups_w [C][G][120], ups_b; src_w [8][G]; alpha [G]; rb_w [G][G][3], rb_b; post_w [22][G][7], post_b */

static float sigm(float v) { return 1.0f / (1.0f + skw_expf(-v)); }
static float leaky(float v, float s) { return v > 0.0f ? v : v * s; }

/* out[t][co] = b[co] + sum_k sum_ci w[co][ci][k] * pre(in[t + k - K/2][ci]) */
static void conv1d(const float* in, long T, int Cin, const float* w, const float* b, int K, int Cout, float* out, int pre, float slope) {
    const int pad = K / 2;
#pragma omp parallel for
    for (long t = 0; t < T; ++t)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) {
                const long tt = t + k - pad; if (tt < 0 || tt >= T) continue;      /* zero padding adds 0.0 exactly */
                for (int ci = 0; ci < Cin; ++ci) { float v = in[tt * Cin + ci]; if (pre) v = leaky(v, slope); acc += (double)w[((long)co * Cin + ci) * K + k] * (double)v; }
            }
            out[t * Cout + co] = (float)(acc + (double)(b ? b[co] : 0.0f));
        }
}
static void ln_row(const float* x, int d, float* mu_o, float* rstd_o) {
    double s = 0.0; for (int c = 0; c < d; ++c) s += (double)x[c]; const double mean = s / d;
    double q = 0.0; for (int c = 0; c < d; ++c) { const double u = (double)x[c] - mean; q += u * u; }
    *mu_o = (float)mean; *rstd_o = (float)(1.0 / sqrt(q / d + 1e-5));
}
static void style_fc(const float* w, const float* b, const float* s, int rows, float* y) {
    for (int r = 0; r < rows; ++r) { double acc = 0.0; for (int j = 0; j < STYLE_; ++j) acc += (double)w[(long)r * STYLE_ + j] * (double)s[j]; y[r] = (float)(acc + (double)b[r]); }
}
static void adain(const float* z, long F, int C, const float* ada, const float* res, float* out) {
    for (int c = 0; c < C; ++c) {
        double s = 0.0; for (long f = 0; f < F; ++f) s += (double)z[f * C + c]; const double mean = s / F;
        double q = 0.0; for (long f = 0; f < F; ++f) { const double u = (double)z[f * C + c] - mean; q += u * u; }
        const float mu = (float)mean, rstd = (float)(1.0 / sqrt(q / F + 1e-5));
        for (long f = 0; f < F; ++f) { float v = ((z[f * C + c] - mu) * rstd) * (1.0f + ada[c]) + ada[C + c];
        v = leaky(v, 0.2f); out[f * C + c] = res ? (res[f * C + c] + v) * 0.70710678118654752f : v; }
    }
}

/* returns the number of samples written to y (5 (120 F - 1)), or -1; dur [T], f0 / en [F], z [F][C], o [P][22] are stage taps (may be NULL) */
long skwo_tts_synth(const skwo_tts_dims* D, const int* ids, const float* style256, const float* const* w, int* dur_out, int* F_out,
                    float* f0_out, float* en_out, float* z_out, float* o_out, float* y, long y_cap) {
    const int T = D->T, d = D->d, K = D->K, C = D->C, G = D->G; int wi = 0;
    const float* s_ac = style256; const float* s_pr = style256 + STYLE_;
    float* x = (float*)malloc(sizeof(float) * (size_t)T * d); float* yb = (float*)malloc(sizeof(float) * (size_t)T * d); float* h = (float*)malloc(sizeof(float) * (size_t)T * d);
    const float* emb = w[wi++];
    for (int t = 0; t < T; ++t) memcpy(x + (long)t * d, emb + (long)ids[t] * d, sizeof(float) * d);
    for (int i = 0; i < D->n_te; ++i) {
        const float* cw = w[wi++]; const float* cb = w[wi++]; const float* ga = w[wi++]; const float* be = w[wi++];
        conv1d(x, T, d, cw, cb, 5, d, yb, 0, 0.0f);
        for (int t = 0; t < T; ++t) { float mu, rstd; ln_row(yb + (long)t * d, d, &mu, &rstd);
        for (int c = 0; c < d; ++c) x[(long)t * d + c] = leaky(((yb[(long)t * d + c] - mu) * rstd) * ga[c] + be[c], 0.2f); }
    }
    float* ada = (float*)malloc(sizeof(float) * 2 * (size_t)(d > C ? d : C));
    { const float* fw = w[wi++]; const float* fb = w[wi++]; style_fc(fw, fb, s_pr, 2 * d, ada); }
    for (int t = 0; t < T; ++t) { float mu, rstd; ln_row(x + (long)t * d, d, &mu, &rstd);
    for (int c = 0; c < d; ++c) h[(long)t * d + c] = ((x[(long)t * d + c] - mu) * rstd) * (1.0f + ada[c]) + ada[d + c]; }
    const float* dw = w[wi++]; const float* db = w[wi++];
    int* dur = (int*)malloc(sizeof(int) * T); long F = 0;
    for (int t = 0; t < T; ++t) {
        double tot = 0.0;
        for (int k = 0; k < K; ++k) { double s = 0.0; for (int c = 0; c < d; ++c) s += (double)dw[(long)k * d + c] * (double)h[(long)t * d + c]; tot += (double)sigm((float)(s + (double)db[k])); }
        const float r = rintf((float)tot * D->scale); dur[t] = r < 1.0f ? 1 : (int)r; if (dur_out) dur_out[t] = dur[t];
    }
    int* tok = (int*)malloc(sizeof(int) * (size_t)D->max_frames);
    for (int t = 0; t < T; ++t) for (int k = 0; k < dur[t] && F < D->max_frames; ++k) tok[F++] = t;
    if (F_out) *F_out = (int)F;
    const long P = F * U_, n_out = HOP_ * (P - 1);
    if (n_out > y_cap) { free(x); free(yb); free(h); free(ada); free(dur); free(tok); return -1; }
    const float* f0w = w[wi++]; const float* f0s = w[wi++]; const float* f0b = w[wi++]; const float* nw = w[wi++]; const float* nb = w[wi++];
    float* f0 = (float*)malloc(sizeof(float) * F); float* en = (float*)malloc(sizeof(float) * F);
    { double sv = 0.0; for (int j = 0; j < STYLE_; ++j) sv += (double)f0s[j] * (double)s_pr[j];
      for (long f = 0; f < F; ++f) { const float* hr = h + (long)tok[f] * d;
      double a = 0.0, e = 0.0; for (int c = 0; c < d; ++c) { a += (double)f0w[c] * (double)hr[c]; e += (double)nw[c] * (double)hr[c]; }
          f0[f] = 60.0f + 340.0f * sigm((float)(a + sv + (double)f0b[0])); en[f] = (float)(e + (double)nb[0]); } }
    if (f0_out) memcpy(f0_out, f0, sizeof(float) * F);
    if (en_out) memcpy(en_out, en, sizeof(float) * F);
    float* u = (float*)malloc(sizeof(float) * (size_t)F * (d + 2));
    for (long f = 0; f < F; ++f) { memcpy(u + f * (d + 2), x + (long)tok[f] * d, sizeof(float) * d); u[f * (d + 2) + d] = f0[f] / 400.0f; u[f * (d + 2) + d + 1] = en[f]; }
    float* z = (float*)malloc(sizeof(float) * (size_t)F * C); float* r = (float*)malloc(sizeof(float) * (size_t)F * C); float* z2 = (float*)malloc(sizeof(float) * (size_t)F * C);
    { const float* ew = w[wi++]; const float* eb = w[wi++]; const float* fw = w[wi++]; const float* fb = w[wi++];
      conv1d(u, F, d + 2, ew, eb, 3, C, r, 0, 0.0f); style_fc(fw, fb, s_ac, 2 * C, ada); adain(r, F, C, ada, NULL, z); }
    for (int i = 0; i < D->n_dec; ++i) {
        const float* cw = w[wi++]; const float* cb = w[wi++]; const float* fw = w[wi++]; const float* fb = w[wi++];
        conv1d(z, F, C, cw, cb, 3, C, r, 0, 0.0f); style_fc(fw, fb, s_ac, 2 * C, ada); adain(r, F, C, ada, z, z2);
        float* tmp = z; z = z2; z2 = tmp;
    }
    if (z_out) memcpy(z_out, z, sizeof(float) * (size_t)F * C);
    const float* upw = w[wi++]; const float* upb = w[wi++]; const float* srcw = w[wi++];
    const float* alpha = w[wi++]; const float* rbw = w[wi++]; const float* rbb = w[wi++]; const float* pw = w[wi++]; const float* pb = w[wi++];
    double* phi = (double*)malloc(sizeof(double) * F); { double a = 0.0; for (long f = 0; f < F; ++f) { phi[f] = a; a += (double)U_ * (double)f0[f] / SUBRATE_; a -= floor(a); } }
    float* g = (float*)malloc(sizeof(float) * (size_t)P * G); float* g1 = (float*)malloc(sizeof(float) * (size_t)P * G); float* g2 = (float*)malloc(sizeof(float) * (size_t)P * G);
#pragma omp parallel for
    for (long p = 0; p < P; ++p) {
        const long f = p / U_; const int uu = (int)(p % U_); float har[H_];
        for (int hh = 0; hh < H_; ++hh) { const double ph = phi[f] + (double)uu * (double)f0[f] / SUBRATE_; double cyc = (double)(hh + 1) * ph; cyc -= floor(cyc);
            har[hh] = ((float)(hh + 1) * f0[f] < 0.5f * (float)SUBRATE_) ? (float)sin(6.283185307179586476925286766559 * cyc) : 0.0f; }
        for (int cg = 0; cg < G; ++cg) {
            double acc = 0.0; for (int c = 0; c < C; ++c) acc += (double)upw[((long)c * G + cg) * U_ + uu] * (double)z[f * C + c];
            double hs = 0.0; for (int hh = 0; hh < H_; ++hh) hs += (double)srcw[hh * G + cg] * (double)har[hh];
            g[p * G + cg] = (float)(acc + hs + (double)upb[cg]);
        }
    }
    for (long i = 0; i < P * G; ++i) { const float a = alpha[i % G]; const float s = sinf(a * g[i]); g1[i] = g[i] + s * s / a; }
    conv1d(g1, P, G, rbw, rbb, 3, G, g2, 0, 0.0f);
    for (long i = 0; i < P * G; ++i) g1[i] = g[i] + g2[i];
    float* o = (float*)malloc(sizeof(float) * (size_t)P * 2 * BINS_);
    conv1d(g1, P, G, pw, pb, 7, 2 * BINS_, o, 1, 0.01f);
    if (o_out) memcpy(o_out, o, sizeof(float) * (size_t)P * 2 * BINS_);
#pragma omp parallel for
    for (long n = 0; n < n_out; ++n) {
        const long pos = n + NFFT_ / 2; double acc = 0.0, wsum = 0.0;
        long p_lo = (pos - (NFFT_ - 1) + HOP_ - 1) / HOP_; if (pos - (NFFT_ - 1) < 0) p_lo = 0; const long p_hi = pos / HOP_;
        for (long p = p_lo; p <= p_hi && p < P; ++p) {
            const int m = (int)(pos - p * HOP_); const double wnd = 0.5 - 0.5 * cos(6.283185307179586476925286766559 * m / NFFT_); const float* op = o + p * (2 * BINS_); double xs = 0.0;
            for (int k = 0; k < BINS_; ++k) {
                const float mag = skw_expf(op[k]); const float ph = sinf(op[BINS_ + k]);
                const double re = (double)mag * cos((double)ph), im = (double)mag * sin((double)ph); const double ang = 6.283185307179586476925286766559 * k * m / NFFT_;
                xs += (k == 0) ? re : (k == BINS_ - 1) ? re * cos(ang) : 2.0 * (re * cos(ang) - im * sin(ang));
            }
            acc += wnd * xs / NFFT_; wsum += wnd * wnd;
        }
        y[n] = wsum > 1e-11 ? (float)(acc / wsum) : 0.0f;
    }
    free(x); free(yb); free(h); free(ada); free(dur); free(tok); free(f0); free(en); free(u); free(z); free(r); free(z2); free(phi); free(g); free(g1); free(g2); free(o);
    return n_out;
}
