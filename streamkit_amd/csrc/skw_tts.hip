// skw_tts.hip — libskw_tts.so: the speech synthesiser behind the Kokoro TTS node (include/skw_tts.h), hand-written HIP for gfx950.
//
// Replaces the sherpa-onnx calls of /root/reference/plugins/native/kokoro/src/ffi.rs:119-137 (create / generate / destroy).
// PARITY UNPINNED (header of include/skw_tts.h): Kokoro-82M's graph and weights are not in /root/reference.  The network evaluated here is the published
// architecture as recalled — ALBERT text encoder, BiLSTM / AdaLayerNorm prosody predictor with AdainResBlk1d F0 / energy branches, acoustic text encoder,
// AdaIN decoder, ISTFTNet generator with a harmonic-plus-noise source — wired in include/skw_kokoro_net.h (shared with the CPU checker,
// oracle/skw_kokoro_oracle.cpp, which implements every operator as a plain loop); this file implements every operator as a HIP kernel.
//
// Arithmetic contract (skw_kokoro_net.h): every weight product is one k-ascending f32 fma chain.  The convolutions and linear layers — all of the network's
// FLOPs — run it on the matrix cores: v_mfma_f32_16x16x4_f32 IS that chain bit for bit (tools/probe/probe_mfma.hip), so conv / linear outputs equal the
// checker's exactly; LSTM recurrences, attention and the small style projections chain on the VALU in the same order.  Statistics and softmax sums are f64.
// What differs between the two backends is the platform's sin / cos / atan2 (Snake, source, STFT): a few ulps, bounded by the tests' waveform tolerance.
#include "../../include/skw_tts.h"
#include "../../include/skw_kokoro_net.h"
#include "../../include/skw_math.h"
#include "skw_silero.h"      // the ONNX initializer reader (skw::onnx)
#include "skw_tts_text.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>

using namespace skw::kokoro;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define TTS_MAX_FRAMES 3000      // 75 s of audio per call (a sentence; the node splits its text: kokoro_node.rs:444-492)

static void set_err(char* err, size_t n, const char* fmt, ...) { if (!err || !n) return; va_list ap; va_start(ap, fmt); vsnprintf(err, n, fmt, ap); va_end(ap); }

// ------------------------------------------------------------------ kernels
__device__ const double TWC[N_FFT] = SKW_KOKORO_TW_COS, TWS[N_FFT] = SKW_KOKORO_TW_SIN;
__device__ __forceinline__ float sigmoid_e(float v) { return 1.0f / (1.0f + skw_expf(-v)); }
__device__ __forceinline__ float tanh_e(float v) { const float e = skw_expf(2.0f * v); return 1.0f - 2.0f / (e + 1.0f); }

// conv / linear on the matrix cores.  out[t * os + oo][co] = bias[co] + chain_{tap, ci} w[co][ci][tap] * x[t * stride + tap * dil - pad][ci]   (os = 1, oo = 0 for a convolution; a
// ConvTranspose1d is `stride` such launches, one per output phase, with dil = -1: see GpuBackend::convtr).
// Weights arrive packed as [k / 4][Co padded to 16][4] with k = tap * Ci + ci (zero padded to a multiple of 4): lane (i = lane & 15, kq = lane >> 4) of the MFMA's first
// operand is one coalesced float.  A workgroup is 64 output channels x 64 time steps, wave w = channel tile w x four 16-step tiles; a lane ends with 4 adjacent
// channels of one step: one 16-byte store.  The gather of x is plain loads (activations are small and L2-resident; the network's weights stream once per launch).
__global__ __launch_bounds__(256)
void k_tts_conv(const float* x, int T, int Ci, const float* wp, int Co, int Co16, const float* bias, int K, int stride, int dil, int pad, int To, float* out, int os, int oo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * 64 + wave * 16, t0 = blockIdx.x * 64;
    if (co0 >= Co16) return;
    const int Ktot = K * Ci, nk4 = (Ktot + 3) >> 2;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tbase[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) tbase[n] = (t0 + 16 * n + i16) * stride - pad;
    int k = kq, tap = 0, ci = kq;
    while (ci >= Ci) { ci -= Ci; ++tap; }
    const float* wrow = wp + ((long)co0 + i16) * 4 + kq;
    for (int k4 = 0; k4 < nk4; ++k4) {
        const float a = wrow[(long)k4 * Co16 * 4];
        const bool kv = k < Ktot;
        const int toff = tap * dil;
        float b[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int tt = tbase[n] + toff; b[n] = (kv && tt >= 0 && tt < T) ? x[(long)tt * Ci + ci] : 0.0f; }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
        k += 4; ci += 4;
        while (ci >= Ci) { ci -= Ci; ++tap; }
    }
    const int co = co0 + 4 * kq;      // this lane's 4 adjacent channels (rows 4 (lane >> 4) + r of the tile)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int t = t0 + 16 * n + i16;
        if (t >= To) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < Co) out[((long)t * os + oo) * Co + co + r] = bias ? acc[n][r] + bias[co + r] : acc[n][r];
    }
}
// The same contraction, tiled for the long launches (the generator's 120 F rows, the decoder's 1090-wide inputs): a workgroup is 16 NT time steps x 64 MT output channels
// (wave w = MT channel tiles x NT 16-step tiles; NT = 8, MT = 2 for the long launches, smaller tiles when the launch would otherwise leave CUs idle).  The 16 NT x 32 slab of the
//  implicit im2col matrix for k = 32 c .. 32 c + 31 is gathered ONCE per workgroup with coalesced loads
// (lanes along k = along ci: 128-byte runs), double-buffered in LDS (row stride 36 floats: the MFMA operand read, 16 rows x 4 k per wave instruction, then touches all 64 banks
// once), and shared by the four waves; weights go from the packed image straight to the first operand as in k_tts_conv.  Same k order, same chain: bit-identical to k_tts_conv.
#define TTS_CT_LDW 36
template <int MT, int NT> __global__ __launch_bounds__(256, 2)
void k_tts_conv_t(const float* x, int T, int Ci, const float* wp, int Co, int Co16, const float* bias, int K, int stride, int dil, int pad, int To, float* out, int os, int oo) {
    constexpr int ROWS = 16 * NT, NG = 2 * NT;      // rows of the slab; rows each staging thread gathers
    __shared__ float tile[2][ROWS * TTS_CT_LDW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * (64 * MT) + wave * (16 * MT), t0 = blockIdx.x * ROWS;
    const int Ktot = K * Ci, nk4 = (Ktot + 3) >> 2, nchunk = (nk4 + 7) >> 3;
    const int s_kk = threadIdx.x & 31, s_r0 = threadIdx.x >> 5;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st[NG];
    auto gather = [&](int c) {      // this thread's column k = 32 c + s_kk of the slab: NG rows
        const int k = 32 * c + s_kk; const bool kv = k < Ktot; const int tap = kv ? k / Ci : 0, ci = k - tap * Ci; const int tb = (t0 + s_r0) * stride + tap * dil - pad;
#pragma unroll
        for (int j = 0; j < NG; ++j) { const int tt = tb + 8 * j * stride; st[j] = (kv && tt >= 0 && tt < T) ? x[(long)tt * Ci + ci] : 0.0f; }
    };
    auto put = [&](int b) {
#pragma unroll
        for (int j = 0; j < NG; ++j) tile[b][(s_r0 + 8 * j) * TTS_CT_LDW + s_kk] = st[j];
    };
    bool mv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mv[m] = co0 + 16 * m < Co16;
    const float* wrow = wp + ((long)co0 + i16) * 4 + kq;
    float an[8][MT], ac[8][MT];
    auto loadw = [&](int c) {      // chunk c's 8 x MT first operands; steps past the end of k multiply zeros (their slab columns are zero too)
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
            for (int m = 0; m < MT; ++m) an[k4][m] = (mv[m] && 8 * c + k4 < nk4) ? wrow[((long)(8 * c + k4) * Co16 + 16 * m) * 4] : 0.0f;
    };
    gather(0); loadw(0); put(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int b = c & 1;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
            for (int m = 0; m < MT; ++m) ac[k4][m] = an[k4][m];
        if (c + 1 < nchunk) { gather(c + 1); loadw(c + 1); }      // the next chunk's slab column and weights are in flight under this chunk's MFMAs
        const float* tb = &tile[b][i16 * TTS_CT_LDW + kq];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            float bb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bb[n] = tb[16 * n * TTS_CT_LDW + 4 * k4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k4][m], bb[n], acc[m][n], 0, 0, 0);
        }
        if (c + 1 < nchunk) put(b ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = co0 + 16 * m + 4 * kq;
        if (co >= Co) continue;
        const bool vec = co + 3 < Co && (Co & 3) == 0;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (bias && co + r < Co) ? bias[co + r] : 0.0f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int t = t0 + 16 * n + i16;
            if (t >= To) continue;
            float* o = out + ((long)t * os + oo) * Co + co;
            if (vec) { *(f32x4*)o = bias ? (f32x4){acc[m][n][0] + bv[0], acc[m][n][1] + bv[1], acc[m][n][2] + bv[2], acc[m][n][3] + bv[3]} : acc[m][n]; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (co + r < Co) o[r] = bias ? acc[m][n][r] + bv[r] : acc[m][n][r];
            }
        }
    }
}
// ConvTranspose1d: out[u][co] = bias[co] + chain_{tap asc (u + pad - tap = t * stride), ci asc} w[ci][co][tap] * x[t][ci]; weights pre-transposed to [tap][ci][co]
__global__ __launch_bounds__(256) void k_tts_convtr(const float* x, int T, int Ci, const float* wt, int Co, const float* bias, int K, int stride, int pad, int To, float* out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x; if (idx >= (long)To * Co) return;
    const int u = (int)(idx / Co), co = (int)(idx % Co);
    float acc = 0.0f;
    for (int tap = 0; tap < K; ++tap) {
        const int num = u + pad - tap; if (num < 0 || num % stride) continue; const int t = num / stride; if (t >= T) continue;
        const float* xr = x + (long)t * Ci; const float* wr = wt + ((long)tap * Ci) * Co + co;
        for (int ci = 0; ci < Ci; ++ci) acc = __builtin_fmaf(wr[(long)ci * Co], xr[ci], acc);
    }
    out[idx] = bias ? acc + bias[co] : acc;
}
// depthwise ConvTranspose1d (the AdainResBlk1d `pool`): w [C][1][K]
__global__ __launch_bounds__(256) void k_tts_convtr_dw(const float* x, int T, int C, const float* w, const float* bias, int K, int stride, int pad, int To, float* out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x; if (idx >= (long)To * C) return;
    const int u = (int)(idx / C), c = (int)(idx % C);
    float acc = 0.0f;
    for (int tap = 0; tap < K; ++tap) { const int num = u + pad - tap; if (num < 0 || num % stride) continue; const int t = num / stride;
    if (t >= T) continue; acc = __builtin_fmaf(w[c * K + tap], x[(long)t * C + c], acc); }
    out[idx] = bias ? acc + bias[c] : acc;
}
__device__ __forceinline__ double tts_block_sum(double v, double* sh) {      // blockDim.x == 256
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads(); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v; __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
// LayerNorm over the channels of row t: mode 0 y = n * gamma + beta, mode 1 (AdaLN) y = n * (1 + gb[c]) + gb[C + c]
__global__ __launch_bounds__(256) void k_tts_ln(float* x, int C, const float* gamma, const float* beta, const float* gb, int mode, float eps) {
    __shared__ double sh[4]; float* xr = x + (long)blockIdx.x * C;
    double s = 0.0; for (int c = threadIdx.x; c < C; c += 256) s += (double)xr[c];
    const double mean = tts_block_sum(s, sh) / C;
    double q = 0.0; for (int c = threadIdx.x; c < C; c += 256) { const double u = (double)xr[c] - mean; q += u * u; }
    const double var = tts_block_sum(q, sh) / C; const float rstd = (float)(1.0 / sqrt(var + (double)eps)); const float mu = (float)mean;
    for (int c = threadIdx.x; c < C; c += 256) { const float n = (xr[c] - mu) * rstd; xr[c] = mode == 0 ? n * gamma[c] + beta[c] : n * (1.0f + gb[c]) + gb[C + c]; }
}
// instance norm over time, per channel: partial f64 sums of x (mean == nullptr) or of (x - mean)^2 over row chunks of 512; thread = channel (coalesced rows)
__global__ __launch_bounds__(64) void k_tts_in_partial(const float* x, int T, int C, const double* mean, double* part) {
    const int c = blockIdx.y * 64 + threadIdx.x; if (c >= C) return;
    const int t0 = blockIdx.x * 512, t1 = min(T, t0 + 512); double s = 0.0; const double m = mean ? mean[c] : 0.0; const bool sq = mean != nullptr;
    int t = t0;
    for (; t + 16 <= t1; t += 16) {      // sixteen loads in flight, then their adds in row order (the chain of f64 adds is the contract; the loads are what it waited for)
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = x[(long)(t + j) * C + c];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const double u = (double)v[j] - m; s += sq ? u * u : u; }
    }
    for (; t < t1; ++t) { const double u = (double)x[(long)t * C + c] - m; s += sq ? u * u : u; }
    part[(long)blockIdx.x * C + c] = s;
}
// chunk sums in ascending order -> mean (stage 0) or (mean as f32, rstd) (stage 1)
__global__ void k_tts_in_final(const double* part, int nchunk, int T, int C, double* mean, float* stats, int stage) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= C) return;
    double s = 0.0; int i = 0;
    for (; i + 8 <= nchunk; i += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(long)(i + j) * C + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; i < nchunk; ++i) s += part[(long)i * C + c];
    if (stage == 0) mean[c] = s / T; else { stats[c] = (float)mean[c]; stats[C + c] = (float)(1.0 / sqrt(s / T + (double)1e-5f)); }
}
// y[r] = bias[r] + chain_j w[r][j] * s[j]   (style projections: 128 inputs)
__global__ void k_tts_style_fc(const float* w, const float* bias, const float* s, int rows, float* y) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= rows) return;
    float acc = 0.0f; for (int j = 0; j < STYLE_DIM; ++j) acc = __builtin_fmaf(w[(long)r * STYLE_DIM + j], s[j], acc);
    y[r] = acc + bias[r];
}
__device__ __forceinline__ float tts_activate(float v, int kind, const float* alpha, int c) {
    if (kind == ACT_LEAKY02) return v > 0.0f ? v : v * 0.2f;
    if (kind == ACT_LEAKY01) return v > 0.0f ? v : v * 0.1f;
    if (kind == ACT_LEAKY001) return v > 0.0f ? v : v * 0.01f;
    if (kind == ACT_GELU) { const float u = 0.79788456080286535588f * (v + 0.044715f * ((v * v) * v)); return (0.5f * v) * (1.0f + tanh_e(u)); }
    const float al = alpha[c]; const float s = sinf(al * v); return v + (s * s) / al;
}
__global__ void k_tts_act(float* x, long n, int C, int kind, const float* alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    x[i] = tts_activate(x[i], kind, alpha, (int)(i % C));
}
// instance-norm affine and the activation that always follows it, one pass over the rows instead of two (the same f32 value goes into the activation either way)
__global__ void k_tts_in_apply_act(float* x, long n, int C, const float* stats, const float* gb, int kind, const float* alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const int c = (int)(i % C);
    const float v = ((x[i] - stats[c]) * stats[C + c]) * (1.0f + gb[c]) + gb[C + c];
    x[i] = tts_activate(v, kind, alpha, c);
}
__global__ void k_tts_add_scale(float* a, const float* b, float f, long n) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = (a[i] + b[i]) * f; }
__global__ void k_tts_add(float* a, const float* b, long n) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] + b[i]; }
__global__ void k_tts_scale(float* a, float f, long n) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * f; }
// dst[t][c0 + c] = src[row(t)][c]: concat pieces, nearest up-sampling (div = 2), row gathers (rows != nullptr), the reflection pad (shift = 1)
__global__ void k_tts_copy_cols(const float* src, int Cs, float* dst, int Cd, int c0, int T, int div, const int* rows, int shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long)T * Cs) return;
    const int t = (int)(i / Cs), c = (int)(i % Cs); int r = rows ? rows[t] : t / div;
    if (shift) r = t == 0 ? 1 : t - 1;
    dst[(long)t * Cd + c0 + c] = src[(long)r * Cs + c];
}
__global__ void k_tts_fill_style(float* dst, int Cd, int c0, int T, const float* s) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long)T * STYLE_DIM) return;
    dst[(long)(i / STYLE_DIM) * Cd + c0 + (i % STYLE_DIM)] = s[i % STYLE_DIM]; }
__global__ void k_tts_embed(const float* emb, const int* ids, int D, float* x, const float* pos, const float* type) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) { const float v = emb[(long)ids[t] * D + c]; x[(long)t * D + c] = pos ? (v + pos[(long)t * D + c]) + type[c] : v; }
}
// self attention of the ALBERT layer: one workgroup per (query i, head h); scores and outputs are k-ascending chains, softmax with skw_expf and an f64 sum
__global__ __launch_bounds__(256) void k_tts_attention(const float* q, const float* k, const float* v, int T, int C, float* out) {
    extern __shared__ float sc[];      // [T] scores -> probabilities
    __shared__ double sh[4]; __shared__ float shm[4];
    const int i = blockIdx.x, h = blockIdx.y; const float* qi = q + (long)i * C + 64 * h;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < T; j += 256) { const float* kj = k + (long)j * C + 64 * h;
    float acc = 0.0f; for (int c = 0; c < 64; ++c) acc = __builtin_fmaf(qi[c], kj[c], acc); acc = acc * 0.125f; sc[j] = acc; m = fmaxf(m, acc); }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
    double s = 0.0;
    for (int j = threadIdx.x; j < T; j += 256) { const float e = skw_expf(sc[j] - m); sc[j] = e; s += (double)e; }
    const float fs = (float)tts_block_sum(s, sh);
    for (int j = threadIdx.x; j < T; j += 256) sc[j] = sc[j] / fs;
    __syncthreads();
    if (threadIdx.x < 64) { const int c = threadIdx.x; float acc = 0.0f;
    for (int j = 0; j < T; ++j) acc = __builtin_fmaf(sc[j], v[(long)j * C + 64 * h + c], acc); out[(long)i * C + 64 * h + c] = acc; }
}
// one direction of an LSTM per workgroup (blockIdx.x = direction): thread gi < 4H chains its gate row over h (LDS) from the transposed recurrent weight [k][4H],
// threads j < H then update c / h.  xp [T][4H] holds W_ih x + b_ih (k_tts_conv); gate order i, f, g, o.
__global__ __launch_bounds__(1024) void k_tts_lstm(const float* xp_f, const float* xp_r, const float* whhT_f, const float* whhT_r, const float* bhh_f, const float* bhh_r, int T, int H, float* out) {
    extern __shared__ float ls[];      // h [H] | a [4H]
    float* hs = ls; float* as = ls + H;
    const int dir = blockIdx.x, gi = threadIdx.x, G4 = 4 * H;
    const float* xp = dir ? xp_r : xp_f; const float* whhT = dir ? whhT_r : whhT_f; const float* bhh = dir ? bhh_r : bhh_f;
    float c = 0.0f;
    if (gi < H) hs[gi] = 0.0f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        if (gi < G4) { float acc = 0.0f; for (int k = 0; k < H; ++k) acc = __builtin_fmaf(whhT[(long)k * G4 + gi], hs[k], acc); as[gi] = (xp[(long)t * G4 + gi] + acc) + bhh[gi]; }
        __syncthreads();
        if (gi < H) {
            const float ig = sigmoid_e(as[gi]), fg = sigmoid_e(as[H + gi]), gg = tanh_e(as[2 * H + gi]), og = sigmoid_e(as[3 * H + gi]);
            c = (fg * c) + (ig * gg); const float hv = og * tanh_e(c);
            hs[gi] = hv; out[(long)t * 2 * H + dir * H + gi] = hv;
        }
        __syncthreads();
    }
}
// The same recurrence spread over H / 32 workgroups per direction, each on its own CU: workgroup j keeps the four gate rows of hidden units 32 j .. 32 j + 31 — its 128 x H slice of
// the recurrent weight — in LDS for the whole sequence (128 KiB at H = 256), so a step reads no weight from L2 (k_tts_lstm re-reads the full 1 MiB per step through ONE CU's load path:
// ~9 us per step).  A step's h slices meet through global memory with ONE round trip each way: a unit's h is stored as a 64-bit word {launch epoch : step + 1, h} (single-copy atomic),
// and the consumers poll the words they need until the tag is this step's — no counter, no fence.  Buffers alternate by step parity (a workgroup cannot run two steps ahead: it needs
// everyone's step s to start s + 1).  Thread = (gate, unit): the same k-ascending chain over h as k_tts_lstm, so results are bit-identical.  Every wait is bounded: a poll that
// outlasts ~1 s raises `err`, and every later wait of every workgroup gives up as soon as it sees the flag.
__global__ __launch_bounds__(128) void k_tts_lstm_mw(const float* xp_f, const float* xp_r, const float* whhT_f, const float* whhT_r, const float* bhh_f, const float* bhh_r, int T, int H, float* out,
                                                     unsigned long long* hbuf, unsigned epoch, int* err) {
    extern __shared__ float ls[];      // w [H / 4][128][4] | h [H] | a [128]
    float* w = ls; float* hs = ls + (size_t)H * 128; float* as = hs + H;
    const int NW = H / 32, dir = blockIdx.x / NW, j = blockIdx.x % NW, tid = threadIdx.x, g = tid >> 5, u = tid & 31, G4 = 4 * H, gi = g * H + 32 * j + u;
    const float* xp = dir ? xp_r : xp_f; const float* whhT = dir ? whhT_r : whhT_f; const float bh = (dir ? bhh_r : bhh_f)[gi];
    for (int k = 0; k < H; ++k) w[((k >> 2) * 128 + tid) * 4 + (k & 3)] = whhT[(long)k * G4 + gi];
    for (int k = tid; k < H; k += 128) hs[k] = 0.0f;
    float c = 0.0f; bool dead = false;
    float xv_next = xp[(long)(dir ? T - 1 : 0) * G4 + gi];      // W_ih x + b_ih of the step after this one is requested a step ahead: its round trip is not on the chain
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        const float xv = xv_next;
        if (s + 1 < T) xv_next = xp[(long)(dir ? t - 1 : t + 1) * G4 + gi];
        // the chain over h, eight 4-k groups at a time with the NEXT eight's LDS reads already in flight (one wave per SIMD has nobody else to hide an LDS round trip behind).
        // Measured: this is not what bounds a step — 4.1 us with or without the prefetch; ~3 us of it are the two device-scope trips of the h exchange (workgroups sit on different XCDs,
        // so the store and the poll both go past the XCD's L2 to the memory side)
        float acc = 0.0f;
        f32x4 wc[8], hc[8], wn[8], hn[8];
        const f32x4* wq = (const f32x4*)w + tid; const f32x4* hq = (const f32x4*)hs;
#pragma unroll
        for (int i = 0; i < 8; ++i) { wc[i] = wq[i * 128]; hc[i] = hq[i]; }
        for (int kb = 0; kb < (H >> 5); ++kb) {
            const int nb = min(kb + 1, (H >> 5) - 1) * 8;      // (the last block re-reads itself: no branch in the loop)
#pragma unroll
            for (int i = 0; i < 8; ++i) { wn[i] = wq[(nb + i) * 128]; hn[i] = hq[nb + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_fmaf(wc[i][0], hc[i][0], acc); acc = __builtin_fmaf(wc[i][1], hc[i][1], acc);
                acc = __builtin_fmaf(wc[i][2], hc[i][2], acc); acc = __builtin_fmaf(wc[i][3], hc[i][3], acc);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { wc[i] = wn[i]; hc[i] = hn[i]; }
        }
        as[tid] = (xv + acc) + bh;
        __syncthreads();
        const unsigned tag = (epoch << 12) | (unsigned)(s + 1);
        unsigned long long* hb = hbuf + (size_t)((s & 1) * 2 + dir) * H;
        if (tid < 32) {
            const float ig = sigmoid_e(as[u]), fg = sigmoid_e(as[32 + u]), gg = tanh_e(as[64 + u]), og = sigmoid_e(as[96 + u]);
            c = (fg * c) + (ig * gg); const float hv = og * tanh_e(c);
            if (s + 1 < T) __hip_atomic_store(&hb[32 * j + u], ((unsigned long long)tag << 32) | __float_as_uint(hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[(long)t * 2 * H + dir * H + 32 * j + u] = hv;
        }
        if (s + 1 < T) {
            __syncthreads();      // a[] is free again; h[] below is not read by anyone until the next barrier
            for (int k = tid; k < H; k += 128) {
                unsigned long long v = __hip_atomic_load(&hb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); long spins = 0;
                while ((unsigned)(v >> 32) != tag && !dead) {
                    if ((++spins & 1023) == 0 && (spins > (1L << 24) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { *err = 1; dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(&hb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                hs[k] = __uint_as_float((unsigned)v);
            }
            __syncthreads();
        }
    }
}
__global__ void k_tts_durations(const float* lg, int T, int K, float scale, int* dur) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= T) return;
    double s = 0.0; for (int k = 0; k < K; ++k) s += (double)sigmoid_e(lg[(long)t * K + k]);
    const float r = rintf((float)s * scale); dur[t] = r < 1.0f ? 1 : (int)r;
}
// source (the published SineGen law in f64, skw_kokoro_net.h header): per harmonic the frame-rate cumulative phase C[h][m] (inclusive, mod 1; one thread per harmonic walks
// M <= 6000 values), then every sample on its own: C interpolated linearly to the sample rate, times the up-sampling factor
__global__ void k_tts_phase_scan(const float* f0, int M, double* C) {
    const int h = threadIdx.x + 1; if (h > N_HARM) return;
    double c = 0.0;
    for (int m = 0; m < M; ++m) { double r = (double)h * (double)f0[m] / (double)SAMPLE_RATE; r -= floor(r); c += r; c -= floor(c); C[(long)(h - 1) * M + m] = c; }
}
__global__ void k_tts_source(const float* f0c, const double* C, int M, long L, const float* lw, const float* lb, float* src) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x; if (n >= L) return;
    const int m = (int)(n / SRC_UP); const float f0 = f0c[m];
    const float uv = f0 > 10.0f ? 1.0f : 0.0f; const float amp = f0 > 10.0f ? 0.003f : 0.1f / 3.0f;
    double x = ((double)n + 0.5) / (double)SRC_UP - 0.5; if (x < 0.0) x = 0.0;
    const int m0 = (int)x, m1 = m0 + 1 < M ? m0 + 1 : M - 1; const double wq = x - (double)m0;
    float acc = 0.0f;
    for (int h = 1; h <= N_HARM; ++h) {
        double r1 = 0.0; if (m1 > m0) { r1 = (double)h * (double)f0c[m1] / (double)SAMPLE_RATE; r1 -= floor(r1); }
        double cyc = (double)SRC_UP * (C[(long)(h - 1) * M + m0] + wq * r1); cyc -= floor(cyc);
        const float sine = (float)sin(6.283185307179586476925286766559 * cyc) * 0.1f;
        const float val = sine * uv + amp * unit_noise((uint64_t)n * 16 + (uint64_t)h);
        acc = __builtin_fmaf(lw[h - 1], val, acc);
    }
    src[n] = tanh_e(acc + lb[0]);
}
__global__ void k_tts_stft(const float* src, long L, int P, float* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long)P * N_BINS) return;
    const int p = (int)(idx / N_BINS), k = (int)(idx % N_BINS); double re = 0.0, im = 0.0;
    for (int mm = 0; mm < N_FFT; ++mm) {
        long i = (long)p * HOP + mm - N_FFT / 2; if (i < 0) i = -i; if (i >= L) i = 2 * (L - 1) - i;
        const double wv = (0.5 - 0.5 * TWC[mm]) * (double)src[i];
        const int j = (k * mm) % N_FFT;
        re += wv * TWC[j]; im -= wv * TWS[j];
    }
    out[(long)p * 2 * N_BINS + k] = (float)sqrt(re * re + im * im); out[(long)p * 2 * N_BINS + N_BINS + k] = (float)atan2(im, re);
}
// inverse STFT: windowed overlap-add of the frames that cover output sample n, normalised by the window energy, centre-trimmed
__global__ void k_tts_istft(const float* o, int P, float* y, long n_out) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x; if (n >= n_out) return;
    const long pos = n + N_FFT / 2; double acc = 0.0, wsum = 0.0;
    long p_lo = (pos - (N_FFT - 1) + HOP - 1) / HOP; if (pos - (N_FFT - 1) < 0) p_lo = 0; const long p_hi = pos / HOP;
    for (long p = p_lo; p <= p_hi && p < P; ++p) {
        const int mm = (int)(pos - p * HOP); const double wnd = 0.5 - 0.5 * TWC[mm]; const float* op = o + p * (2 * N_BINS); double xs = 0.0;
        for (int k = 0; k < N_BINS; ++k) {
            const float mag = skw_expf(op[k]); const float ph = sinf(op[N_BINS + k]);
            const double re = (double)mag * cos((double)ph), im = (double)mag * sin((double)ph); const int j = (k * mm) % N_FFT;
            xs += (k == 0) ? re : (k == N_BINS - 1) ? re * TWC[j] : 2.0 * (re * TWC[j] - im * TWS[j]);      // bins 0 and N/2 are real in an inverse real DFT
        }
        acc += wnd * xs / N_FFT; wsum += wnd * wnd;
    }
    y[n] = wsum > 1e-11 ? (float)(acc / wsum) : 0.0f;
}

// ------------------------------------------------------------------ engine
static int g_conv_mode = 0;      // skw_tts_debug_conv_mode: 0 automatic, 1 the untiled kernel, 2 the tiled one
static int g_lstm_mode = 0;      // skw_tts_debug_lstm_mode: 0 automatic, 1 one workgroup per direction, 2 H / 32 workgroups per direction
extern "C" void skw_tts_debug_conv_mode(int mode) { g_conv_mode = mode; }
extern "C" void skw_tts_debug_lstm_mode(int mode) { g_lstm_mode = mode; }
struct skw_tts {
    int device = 0; hipStream_t stream = nullptr; std::mutex mu; char errbuf[512] = {0};
    Weights w; Dims g; std::vector<void*> allocs; float length_scale = 1.0f;
    float* voices = nullptr; int n_spk = 0, voice_rows = 0;
    TtsText text;
    // scratch arena: chunks kept for the engine's life, bump-allocated per call (a call's buffers are all live until it ends)
    struct Chunk { char* p; size_t cap; }; std::vector<Chunk> chunks; size_t cur_chunk = 0, cur_off = 0; bool arena_failed = false;
    unsigned long long* lstm_sync = nullptr; unsigned lstm_epoch = 0;      // [4 H] tagged h exchange words (two step parities x two directions) | [1] error flag (low half of a word)
    bool taps_on = false; std::vector<float> dbg[9]; float last_ms = 0.0f;
};
static void* dev_upload(skw_tts* t, const void* h, size_t bytes) {
    void* d = nullptr; if (hipMalloc(&d, std::max<size_t>(16, bytes)) != hipSuccess) return nullptr;
    if (bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    t->allocs.push_back(d); return d;
}
// tests (SKW_TEST_ALLOC_POISON=1, tests/conftest.py): arena chunks start as NaNs (every activation buffer of the synthesiser is a float buffer carved from them)
static int g_tts_alloc_poison = 0;
extern "C" void skw_tts_debug_alloc_poison(int on) { g_tts_alloc_poison = on; }
static void* arena_get(skw_tts* t, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    while (t->cur_chunk < t->chunks.size() && t->cur_off + bytes > t->chunks[t->cur_chunk].cap) { ++t->cur_chunk; t->cur_off = 0; }
    if (t->cur_chunk == t->chunks.size()) {
        const size_t cap = std::max<size_t>(bytes, (size_t)256 << 20); char* p = nullptr;
        if (hipMalloc((void**)&p, cap) != hipSuccess) { t->arena_failed = true; return nullptr; }
        if (g_tts_alloc_poison) (void)hipMemset(p, 0xFF, cap);
        t->chunks.push_back({p, cap}); t->cur_off = 0;
    }
    void* r = t->chunks[t->cur_chunk].p + t->cur_off; t->cur_off += bytes; return r;
}

// the GPU backend of skw::kokoro::Net: every operator launches kernels on the engine's stream into arena buffers
struct GpuBackend {
    struct Buf { float* p = nullptr; int T = 0, C = 0; };
    skw_tts* t; hipStream_t s;
    explicit GpuBackend(skw_tts* t_) : t(t_), s(t_->stream) {}
    Buf make(int T, int C) { Buf b; b.T = T; b.C = C; b.p = (float*)arena_get(t, sizeof(float) * (size_t)std::max(1, T) * C); return b; }
    static unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }
    const float* dev(const Tensor& w) { return (const float*)w.dev; }
    Buf copy(const Buf& x) { Buf b = make(x.T, x.C); if (b.p) hipMemcpyAsync(b.p, x.p, sizeof(float) * (size_t)x.T * x.C, hipMemcpyDeviceToDevice, s); return b; }
    int* upload_ints(const int* v, int n) { int* d = (int*)arena_get(t, sizeof(int) * (size_t)n); if (d) hipMemcpyAsync(d, v, sizeof(int) * n, hipMemcpyHostToDevice, s); return d; }
    Buf embed(const Tensor& tab, const int* ids, int T) {
        Buf b = make(T, (int)tab.dims[1]); int* d = upload_ints(ids, T);
        if (b.p && d) hipLaunchKernelGGL(k_tts_embed, dim3(T), dim3(128), 0, s, dev(tab), d, b.C, b.p, (const float*)nullptr, (const float*)nullptr);
        last_ids = d; return b;
    }
    int* last_ids = nullptr;
    Buf add_pos_type(const Buf& x, const Tensor& pos, const Tensor& type) {      // (re-gathers the word embedding with the position and token-type rows added: (v + pos) + type)
        Buf b = make(x.T, x.C);
        if (b.p) hipLaunchKernelGGL(k_tts_embed, dim3(x.T), dim3(128), 0, s, word_tab, last_ids, x.C, b.p, dev(pos), dev(type));
        return b;
    }
    const float* word_tab = nullptr;
    // g_conv_mode (skw_tts_debug_conv_mode): 1 forces the untiled kernel, 2 the tiled one — both evaluate the same chain and the tests run one against the other;
    // 0 = tiled when the launch is long enough to fill tiles
    template <int MT, int NT> void launch_conv_t(const float* x, int T, int Ci, const float* wp, int Co, int Co16, const float* bias, int K, int stride, int dil,
        int pad, int To, float* out, int os, int oo) {
        hipLaunchKernelGGL((k_tts_conv_t<MT, NT>), dim3((To + 16 * NT - 1) / (16 * NT), (Co16 + 64 * MT - 1) / (64 * MT)), dim3(256), 0, s, x, T, Ci, wp, Co, Co16, bias,
            K, stride, dil, pad, To, out, os, oo);
    }
    void launch_conv(const float* x, int T, int Ci, const float* wp, int Co, const float* bias, int K, int stride, int dil, int pad, int To, float* out, int os = 1, int oo = 0) {
        const int Co16 = (Co + 15) & ~15, mode = g_conv_mode;
        const bool tiled = mode == 2 || (mode != 1 && To >= 32 && (long)K * Ci >= 32);
        if (!tiled) { hipLaunchKernelGGL(k_tts_conv, dim3((To + 63) / 64, (Co16 + 63) / 64), dim3(256), 0, s, x, T, Ci, wp, Co, Co16, bias, K, stride, dil, pad, To, out, os, oo); return; }
        // the largest tile that still gives the 256 CUs a workgroup each (the chain does not allow splitting k, so a short launch can only spread by smaller tiles)
        auto wgs = [&](int mt, int nt) { return (long)((To + 16 * nt - 1) / (16 * nt)) * ((Co16 + 64 * mt - 1) / (64 * mt)); };
        if (Co16 > 64 && wgs(2, 8) >= 256) launch_conv_t<2, 8>(x, T, Ci, wp, Co, Co16, bias, K, stride, dil, pad, To, out, os, oo);
        else if (wgs(1, 8) >= 256) launch_conv_t<1, 8>(x, T, Ci, wp, Co, Co16, bias, K, stride, dil, pad, To, out, os, oo);
        else if (wgs(1, 4) >= 256) launch_conv_t<1, 4>(x, T, Ci, wp, Co, Co16, bias, K, stride, dil, pad, To, out, os, oo);
        else launch_conv_t<1, 2>(x, T, Ci, wp, Co, Co16, bias, K, stride, dil, pad, To, out, os, oo);
    }
    Buf conv(const Buf& x, const Tensor& w, const Tensor* bias, int K, int stride, int dil, int pad) {
        const int Co = (int)w.dims[0], Ci = (int)w.dims[1], To = (x.T + 2 * pad - dil * (K - 1) - 1) / stride + 1;
        Buf o = make(To, Co);
        if (o.p) launch_conv(x.p, x.T, Ci, (const float*)w.packed, Co, bias ? dev(*bias) : nullptr, K, stride, dil, pad, To, o.p);
        return o;
    }
    // ConvTranspose1d.  Output u takes the taps with (u + pad - tap) % stride == 0, ascending — for phase f = (u + pad) % stride that is tap = f + m stride with x row q - m,
    // q = (u + pad) / stride: a convolution over q with K / stride taps walking BACKWARDS (dil = -1), writing every stride-th output row.  So the matrix cores run it as `stride`
    // launches of the convolution kernel on per-phase weight images (upload_weights: [m * Ci + ci] from w[ci][co][f + m stride]); the chain order (tap, then ci) is unchanged.
    Buf convtr(const Buf& x, const Tensor& w, const Tensor* bias, int K, int stride, int pad, int out_pad, bool depthwise) {
        const int Ci = (int)w.dims[0], Co = depthwise ? Ci : (int)w.dims[1], To = (x.T - 1) * stride - 2 * pad + K + out_pad;
        Buf o = make(To, Co);
        if (!o.p) return o;
        const float* bp = bias ? dev(*bias) : nullptr;
        if (depthwise) { hipLaunchKernelGGL(k_tts_convtr_dw, dim3(blocks((long)To * Co)), dim3(256), 0, s, x.p, x.T, Ci, dev(w), bp, K, stride, pad, To, o.p); return o; }
        if (g_conv_mode == 1 || !w.packed2 || K != 2 * stride) {
            hipLaunchKernelGGL(k_tts_convtr, dim3(blocks((long)To * Co)), dim3(256), 0, s, x.p, x.T, Ci, (const float*)w.packed, Co, bp, K, stride, pad, To, o.p);
            return o;
        }
        const int Co16 = (Co + 15) & ~15, taps = K / stride; const size_t img = (size_t)((taps * Ci + 3) / 4) * Co16 * 4;
        for (int f = 0; f < stride; ++f) {
            const int q0 = (pad - f + stride - 1) / stride > 0 ? (pad - f + stride - 1) / stride : 0;      // first q with u = q stride + f - pad >= 0
            const int u0 = q0 * stride + f - pad; if (u0 >= To) continue;
            const int rows = (To - 1 - u0) / stride + 1;
            launch_conv(x.p, x.T, Ci, (const float*)w.packed2 + (size_t)f * img, Co, bp, taps, 1, -1, -q0, rows, o.p, stride, u0);
        }
        return o;
    }
    void layernorm(Buf& x, const Tensor& g, const Tensor& b, float eps) { hipLaunchKernelGGL(k_tts_ln, dim3(x.T), dim3(256), 0, s, x.p, x.C, dev(g), dev(b), (const float*)nullptr, 0, eps); }
    Buf style_fc(const Tensor& w, const Tensor& b, const float* style) {
        const int R = (int)w.dims[0]; Buf o = make(1, R);
        if (o.p) hipLaunchKernelGGL(k_tts_style_fc, dim3((R + 63) / 64), dim3(64), 0, s, dev(w), dev(b), style, R, o.p);
        return o;
    }
    void ada_ln(Buf& x, const Buf& gb) { hipLaunchKernelGGL(k_tts_ln, dim3(x.T), dim3(256), 0, s, x.p, x.C, (const float*)nullptr, (const float*)nullptr, gb.p, 1, 1e-5f); }
    void ada_in_act(Buf& x, const Buf& gb, Act a, const Tensor* alpha) {
        const int nchunk = (x.T + 511) / 512; const dim3 grid(nchunk, (x.C + 63) / 64);
        double* part = (double*)arena_get(t, sizeof(double) * (size_t)nchunk * x.C);
        double* mean = (double*)arena_get(t, sizeof(double) * x.C); float* stats = (float*)arena_get(t, sizeof(float) * 2 * x.C);
        if (!part || !mean || !stats) return;
        hipLaunchKernelGGL(k_tts_in_partial, grid, dim3(64), 0, s, x.p, x.T, x.C, (const double*)nullptr, part);
        hipLaunchKernelGGL(k_tts_in_final, dim3((x.C + 255) / 256), dim3(256), 0, s, part, nchunk, x.T, x.C, mean, stats, 0);
        hipLaunchKernelGGL(k_tts_in_partial, grid, dim3(64), 0, s, x.p, x.T, x.C, (const double*)mean, part);
        hipLaunchKernelGGL(k_tts_in_final, dim3((x.C + 255) / 256), dim3(256), 0, s, part, nchunk, x.T, x.C, mean, stats, 1);
        const long n = (long)x.T * x.C;
        hipLaunchKernelGGL(k_tts_in_apply_act, dim3(blocks(n)), dim3(256), 0, s, x.p, n, x.C, stats, gb.p, (int)a, alpha ? dev(*alpha) : nullptr);
    }
    void act(Buf& x, Act a, const Tensor* alpha) { const long n = (long)x.T * x.C;
    hipLaunchKernelGGL(k_tts_act, dim3(blocks(n)), dim3(256), 0, s, x.p, n, x.C, (int)a, alpha ? dev(*alpha) : nullptr); }
    void put_cols(const Buf& src, Buf& dst, int c0, int div, const int* rows, int shift) {
        hipLaunchKernelGGL(k_tts_copy_cols, dim3(blocks((long)dst.T * src.C)), dim3(256), 0, s, src.p, src.C, dst.p, dst.C, c0, dst.T, div, rows, shift);
    }
    Buf concat(const std::vector<Buf>& parts) {
        int C = 0; for (auto& p : parts) C += p.C; Buf o = make(parts[0].T, C); if (!o.p) return o;
        int c0 = 0; for (auto& p : parts) { put_cols(p, o, c0, 1, nullptr, 0); c0 += p.C; }
        return o;
    }
    Buf concat_style(const Buf& x, const float* style) {
        Buf o = make(x.T, x.C + STYLE_DIM); if (!o.p) return o;
        put_cols(x, o, 0, 1, nullptr, 0);
        hipLaunchKernelGGL(k_tts_fill_style, dim3(blocks((long)x.T * STYLE_DIM)), dim3(256), 0, s, o.p, o.C, x.C, x.T, style);
        return o;
    }
    void add(Buf& a, const Buf& b) { const long n = (long)a.T * a.C; hipLaunchKernelGGL(k_tts_add, dim3(blocks(n)), dim3(256), 0, s, a.p, b.p, n); }
    void add_scale(Buf& a, const Buf& b, float f) { const long n = (long)a.T * a.C; hipLaunchKernelGGL(k_tts_add_scale, dim3(blocks(n)), dim3(256), 0, s, a.p, b.p, f, n); }
    void scale(Buf& a, float f) { const long n = (long)a.T * a.C; hipLaunchKernelGGL(k_tts_scale, dim3(blocks(n)), dim3(256), 0, s, a.p, f, n); }
    Buf upsample2(const Buf& x) { Buf o = make(2 * x.T, x.C); if (o.p) put_cols(x, o, 0, 2, nullptr, 0); return o; }
    Buf reflect_pad_left(const Buf& x) { Buf o = make(x.T + 1, x.C); if (o.p) put_cols(x, o, 0, 1, nullptr, 1); return o; }
    Buf attention(const Buf& q, const Buf& k, const Buf& v, int heads) {
        Buf o = make(q.T, q.C);
        if (o.p) hipLaunchKernelGGL(k_tts_attention, dim3(q.T, heads), dim3(256), sizeof(float) * q.T, s, q.p, k.p, v.p, q.T, q.C, o.p);
        return o;
    }
    Buf lstm_bi(const Buf& x, const Tensor* const* ws) {
        const int H = (int)ws[1]->dims[1]; Buf o = make(x.T, 2 * H);
        // W_ih x + b_ih for every step at once, on the matrix cores (the [4H][In] weight viewed as a k = 1 convolution)
        Buf xp[2];
        for (int dir = 0; dir < 2; ++dir) {
            const Tensor& wih = *ws[4 * dir]; const int G4 = (int)wih.dims[0], In = (int)wih.dims[1];
            xp[dir] = make(x.T, G4);
            if (xp[dir].p) launch_conv(x.p, x.T, In, (const float*)wih.packed, G4, dev(*ws[4 * dir + 2]), 1, 1, 1, 0, x.T, xp[dir].p);
        }
        if (!o.p || !xp[0].p || !xp[1].p) return o;
        const bool multi = H % 32 == 0 && x.T < 4096 && (g_lstm_mode == 2 || (g_lstm_mode != 1 && x.T >= 8));
        if (!multi)
            hipLaunchKernelGGL(k_tts_lstm, dim3(2), dim3(1024), sizeof(float) * 5 * H, s, xp[0].p, xp[1].p, (const float*)ws[1]->packed, (const float*)ws[5]->packed,
                dev(*ws[3]), dev(*ws[7]), x.T, H, o.p);
        else {
            t->lstm_epoch = (t->lstm_epoch + 1) & 0xFFFFFu;      // 20 bits of launch epoch + 12 bits of step in a word's tag (T <= TTS_MAX_FRAMES < 4096)
            if (t->lstm_epoch == 0) { hipMemsetAsync(t->lstm_sync, 0, sizeof(unsigned long long) * 4 * t->g.H, s); t->lstm_epoch = 1; }
            hipLaunchKernelGGL(k_tts_lstm_mw, dim3(2 * (H / 32)), dim3(128), sizeof(float) * ((size_t)H * 128 + H + 128), s, xp[0].p, xp[1].p, (const float*)ws[1]->packed,
                (const float*)ws[5]->packed, dev(*ws[3]), dev(*ws[7]), x.T, H, o.p, t->lstm_sync, t->lstm_epoch, (int*)(t->lstm_sync + 4 * t->g.H));
        }
        return o;
    }
    Buf gather_rows(const Buf& x, const std::vector<int>& rows) {
        Buf o = make((int)rows.size(), x.C); int* d = upload_ints(rows.data(), (int)rows.size());
        if (o.p && d) put_cols(x, o, 0, 1, d, 0);
        return o;
    }
    std::vector<int> durations(const Buf& lg, float scale) {
        std::vector<int> d(lg.T, 1); int* dd = (int*)arena_get(t, sizeof(int) * lg.T); if (!dd) return d;
        hipLaunchKernelGGL(k_tts_durations, dim3((lg.T + 63) / 64), dim3(64), 0, s, lg.p, lg.T, lg.C, scale, dd);
        hipMemcpyAsync(d.data(), dd, sizeof(int) * lg.T, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        return d;
    }
    Buf source_stft(const Buf& f0c, const Tensor& lw, const Tensor& lb) {
        const int M = f0c.T; const long L = (long)M * SRC_UP; const int P = (int)(L / HOP) + 1;
        double* phi = (double*)arena_get(t, sizeof(double) * M * N_HARM); float* src = (float*)arena_get(t, sizeof(float) * L); Buf o = make(P, 2 * N_BINS);
        if (!phi || !src || !o.p) return o;
        hipLaunchKernelGGL(k_tts_phase_scan, dim3(1), dim3(64), 0, s, f0c.p, M, phi);
        hipLaunchKernelGGL(k_tts_source, dim3(blocks(L)), dim3(256), 0, s, f0c.p, phi, M, L, dev(lw), dev(lb), src);
        hipLaunchKernelGGL(k_tts_stft, dim3(blocks((long)P * N_BINS)), dim3(256), 0, s, src, L, P, o.p);
        return o;
    }
    Buf istft(const Buf& post) {
        const long n_out = (long)(post.T - 1) * HOP; Buf y = make((int)n_out, 1);
        if (y.p) hipLaunchKernelGGL(k_tts_istft, dim3(blocks(n_out)), dim3(256), 0, s, post.p, post.T, y.p, n_out);
        return y;
    }
};

// weights to the device: every tensor as it is, plus the image its consumer reads — conv / linear weights [Co][Ci][K] as the MFMA's first operand
// [k / 4][Co padded to 16][4] (k = tap * Ci + ci), ConvTranspose1d weights [Ci][Co][K] as [tap][ci][co], recurrent LSTM weights [4H][H] transposed
static bool upload_weights(skw_tts* t, std::string* err) {
    for (auto& kv : t->w) {
        Tensor& w = kv.second; const std::string& n = kv.first;
        w.dev = dev_upload(t, w.host.data(), w.host.size() * sizeof(float));
        if (!w.dev) { *err = "device allocation failed for '" + n + "'"; return false; }
        const bool is_tr = n.find("generator.ups.") != std::string::npos && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
        const bool is_hh = n.find("weight_hh_l0") != std::string::npos;
        const bool is_pool = n.find("pool.weight") != std::string::npos;
        std::vector<float> img;
        if (is_tr && w.dims[2] % 2 == 0) {      // the matrix-core form: one image per output phase, stride = K / 2 (Kokoro's up-sampling layers: k = 2 x stride)
            const int64_t Ci = w.dims[0], Co = w.dims[1], K = w.dims[2], st = K / 2, Co16 = (Co + 15) & ~15, nk4 = (2 * Ci + 3) / 4;
            std::vector<float> ph((size_t)(st * nk4 * Co16 * 4), 0.0f);
            for (int64_t f = 0; f < st; ++f) for (int64_t m = 0; m < 2; ++m) for (int64_t ci = 0; ci < Ci; ++ci) for (int64_t co = 0; co < Co; ++co) {
                const int64_t kk = m * Ci + ci; ph[(size_t)(((f * nk4 + (kk >> 2)) * Co16 + co) * 4 + (kk & 3))] = w.host[(size_t)((ci * Co + co) * K + f + m * st)]; }
            w.packed2 = dev_upload(t, ph.data(), ph.size() * sizeof(float));
            if (!w.packed2) { *err = "device allocation failed for '" + n + "'"; return false; }
        }
        if (is_tr) {
            const int64_t Ci = w.dims[0], Co = w.dims[1], K = w.dims[2]; img.resize(w.host.size());
            for (int64_t ci = 0; ci < Ci; ++ci) for (int64_t co = 0; co < Co; ++co) for (int64_t k = 0; k < K; ++k) img[(size_t)((k * Ci + ci) * Co + co)] = w.host[(size_t)((ci * Co + co) * K + k)];
        } else if (is_hh) {
            const int64_t G4 = w.dims[0], H = w.dims[1]; img.resize(w.host.size());
            for (int64_t gi = 0; gi < G4; ++gi) for (int64_t k = 0; k < H; ++k) img[(size_t)(k * G4 + gi)] = w.host[(size_t)(gi * H + k)];
        } else if (!is_pool && (w.dims.size() == 3 || (w.dims.size() == 2 && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0
            && n.find("embeddings.") == std::string::npos && n.find(".embedding.") == std::string::npos) || n.find("weight_ih_l0") != std::string::npos)) {
            const int64_t Co = w.dims[0], Ci = w.dims[1], K = w.dims.size() == 3 ? w.dims[2] : 1, Co16 = (Co + 15) & ~15, Kt = K * Ci, nk4 = (Kt + 3) / 4;
            img.assign((size_t)(nk4 * Co16 * 4), 0.0f);
            for (int64_t co = 0; co < Co; ++co) for (int64_t ci = 0; ci < Ci; ++ci) for (int64_t k = 0; k < K; ++k) {
                const int64_t kk = k * Ci + ci; img[(size_t)(((kk >> 2) * Co16 + co) * 4 + (kk & 3))] = w.host[(size_t)((co * Ci + ci) * K + k)]; }
        }
        if (!img.empty()) { w.packed = dev_upload(t, img.data(), img.size() * sizeof(float)); if (!w.packed) { *err = "device allocation failed for '" + n + "'"; return false; } }
    }
    return true;
}

static skw_tts* create_impl(const skw_tts_config* cfg, char* err, size_t errlen) {
    if (!cfg || !cfg->model || !cfg->voices || !cfg->tokens) { set_err(err, errlen, "skw_tts_create: model, voices and tokens paths are required"); return nullptr; }
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (ndev <= 0) { set_err(err, errlen, "no HIP device available: libskw_tts requires an MI355X (gfx950); there is no CPU fallback"); return nullptr; }
    if (cfg->gpu_device < 0 || cfg->gpu_device >= ndev || hipSetDevice(cfg->gpu_device) != hipSuccess) { set_err(err, errlen, "gpu_device %d out of range (%d devices)", cfg->gpu_device, ndev);
    return nullptr; }
    skw_tts* t = new skw_tts(); t->device = cfg->gpu_device; t->length_scale = cfg->length_scale > 0.0f ? cfg->length_scale : 1.0f;
    auto fail = [&](const std::string& m) -> skw_tts* { set_err(err, errlen, "%s", m.c_str()); skw_tts_destroy(t); return nullptr; };
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) return fail("stream creation failed");
    std::string e;
    if (!load_tokens(&t->text, cfg->tokens, &e)) return fail(e);
    load_lexicon(&t->text, cfg->lexicon);
    std::vector<uint8_t> bytes; if (!read_file(cfg->model, &bytes, (size_t)2048 << 20)) return fail(std::string("cannot read model file ") + cfg->model);
    skw::onnx::Model m; if (!skw::onnx::parse_model(bytes, &m, &e)) return fail(std::string("model file ") + cfg->model + ": " + e);
    for (const auto& x : m.tensors) if (!x.data.empty() && !x.name.empty()) { Tensor tt; tt.dims = x.dims; tt.host = x.data; t->w[x.name] = std::move(tt); }
    bytes.clear(); bytes.shrink_to_fit();
    if (!check(t->w, &t->g, &e)) return fail(std::string(cfg->model) + ": " + e);
    if (t->g.H > 256) return fail("model file: LSTM width above 256 per direction is not supported (one workgroup holds a direction's gates)");
    for (auto& kv : t->text.sym2id) if (kv.second < 0 || kv.second >= t->g.n_sym) return fail("tokens file names an id outside the embedding table");
    if (!upload_weights(t, &e)) return fail(e);
    {   // the multi-workgroup LSTM's exchange buffers, and its LDS size (a 128 x H weight slice: over the 64 KiB default)
        std::vector<unsigned long long> z((size_t)4 * t->g.H + 1, 0ull);
        t->lstm_sync = (unsigned long long*)dev_upload(t, z.data(), z.size() * sizeof(unsigned long long));
        if (!t->lstm_sync) return fail("device allocation failed for the LSTM exchange buffers");
        if (hipFuncSetAttribute((const void*)k_tts_lstm_mw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ((size_t)t->g.H * 128 + t->g.H + 128))) != hipSuccess)
            return fail("hipFuncSetAttribute failed for the LSTM kernel");
    }
    {   // voices.bin: f32 [n_spk][rows][256]; rows = 510 in Kokoro's files
        std::vector<uint8_t> vb; if (!read_file(cfg->voices, &vb, (size_t)1024 << 20)) return fail(std::string("cannot read voices file ") + cfg->voices);
        const size_t row = 2 * STYLE_DIM * 4; if (vb.size() < row || vb.size() % row) return fail("voices file: size is not a multiple of 256 floats");
        const size_t rows = vb.size() / row; t->voice_rows = rows % MAX_TOKENS == 0 ? MAX_TOKENS : 1; t->n_spk = (int)(rows / t->voice_rows);
        t->voices = (float*)dev_upload(t, vb.data(), vb.size()); if (!t->voices) return fail("device allocation failed for the voices");
    }
    hipDeviceSynchronize();
    return t;
}
extern "C" skw_tts* skw_tts_create(const skw_tts_config* cfg, char* err, size_t errlen) {
    try { return create_impl(cfg, err, errlen); } catch (const std::exception& e) { set_err(err, errlen, "skw_tts_create: %s", e.what()); return nullptr; }
}
extern "C" void skw_tts_destroy(skw_tts* t) {
    if (!t) return; hipSetDevice(t->device);
    if (t->stream) { hipStreamSynchronize(t->stream); hipStreamDestroy(t->stream); }
    for (void* p : t->allocs) hipFree(p);
    for (auto& c : t->chunks) hipFree(c.p);
    delete t;
}
extern "C" const char* skw_tts_last_error(const skw_tts* t) { return t->errbuf; }
extern "C" int32_t skw_tts_num_speakers(const skw_tts* t) { return t->n_spk; }
extern "C" int32_t skw_tts_sample_rate(const skw_tts*) { return SAMPLE_RATE; }
extern "C" float skw_tts_last_ms(const skw_tts* t) { return t->last_ms; }
extern "C" void skw_tts_debug_enable(skw_tts* t, int on) { std::lock_guard<std::mutex> l(t->mu); t->taps_on = on != 0; }
extern "C" int32_t skw_tts_tokenize(skw_tts* t, const char* text, int32_t* ids, int32_t cap) {
    try { std::vector<int> v = tokenize(&t->text, text ? text : ""); const int n = std::min((int)v.size(), (int)cap); for (int i = 0; i < n; ++i) ids[i] = v[i]; return n; } catch (...) { return -1; }
}
extern "C" long skw_tts_debug_get(skw_tts* t, int what, float* out, long cap) {
    if (what < 0 || what > 8) return -1; std::lock_guard<std::mutex> l(t->mu);
    const auto& v = t->dbg[what]; if (out) memcpy(out, v.data(), sizeof(float) * std::min<long>(cap, (long)v.size())); return (long)v.size();
}

struct EventPair {      // (destroyed on every return path)
    hipEvent_t a = nullptr, b = nullptr;
    bool create() { return hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
};

static const skw_tts_audio* generate_impl(skw_tts* t, const char* text, int32_t sid, float speed, const int32_t* ids_in, int n_ids) {
    std::lock_guard<std::mutex> l(t->mu); t->errbuf[0] = 0;
    auto fail = [&](const char* fmt, ...) -> const skw_tts_audio* { va_list ap; va_start(ap, fmt); vsnprintf(t->errbuf, sizeof t->errbuf, fmt, ap); va_end(ap); return nullptr; };
    if (!text && !ids_in) return fail("null text");
    if (!(speed > 0.0f) || !std::isfinite(speed)) return fail("speed must be positive");
    if (sid < 0 || sid >= t->n_spk) return fail("speaker id %d outside [0, %d)", sid, t->n_spk);
    if (hipSetDevice(t->device) != hipSuccess) return fail("hipSetDevice failed");
    std::vector<int> ids = ids_in ? std::vector<int>(ids_in, ids_in + n_ids) : tokenize(&t->text, text);
    const int T = (int)ids.size();
    if (T <= 2) return fail("no symbol of the text is in the model's token table");
    EventPair ev; if (!ev.create()) return fail("event creation failed");
    t->cur_chunk = 0; t->cur_off = 0; t->arena_failed = false;
    hipEventRecord(ev.a, t->stream);
    // Kokoro's voices are indexed by length: the published pipeline takes pack[len(phonemes) - 1], and the ids carry the pad id at both ends (T = len + 2): row T - 3
    // (rounds 3-4 took T - 2).  What sherpa-onnx's front end picks for the same text is unpinned.
    const int row = std::min(T - 3, t->voice_rows - 1);
    const float* style = t->voices + ((size_t)sid * t->voice_rows + (size_t)std::max(0, row)) * 2 * STYLE_DIM;
    GpuBackend be(t); be.word_tab = (const float*)t->w.at("bert.embeddings.word_embeddings.weight").dev;
    Net<GpuBackend> net(be, t->w, t->g); Outputs<GpuBackend::Buf> out; std::string e;
    if (!net.forward(ids, style, t->length_scale / speed, TTS_MAX_FRAMES, &out, &e)) { hipStreamSynchronize(t->stream); return fail("%s", e.c_str()); }
    hipEventRecord(ev.b, t->stream);
    if (t->arena_failed || !out.audio.p) { hipStreamSynchronize(t->stream); return fail("device allocation failed for the synthesis workspace"); }
    const long n_out = out.audio.T;
    float* host = (float*)malloc(sizeof(float) * (size_t)std::max<long>(1, n_out)); if (!host) return fail("out of memory");
    if (hipMemcpyAsync(host, out.audio.p, sizeof(float) * (size_t)n_out, hipMemcpyDeviceToHost, t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess
        || hipGetLastError() != hipSuccess) {
        free(host); return fail("synthesis kernels failed"); }
    hipEventElapsedTime(&t->last_ms, ev.a, ev.b);
    {   int lstm_err = 0; hipMemcpy(&lstm_err, t->lstm_sync + 4 * t->g.H, sizeof(int), hipMemcpyDeviceToHost);
        if (lstm_err) { hipMemset(t->lstm_sync + 4 * t->g.H, 0, sizeof(int)); free(host); return fail("LSTM workgroups timed out waiting for each other (GPU oversubscribed?)"); } }
    if (t->taps_on) {      // stage taps for the parity tests (skw_tts_debug_enable): 0 durations, 1 F0, 2 N, 3 decoder output, 4 spectrum + phase, 5 bert, 6 d_en, 7 t_en, 8 source STFT
        auto grab = [&](int k, const GpuBackend::Buf& b) { t->dbg[k].resize((size_t)b.T * b.C); hipMemcpy(t->dbg[k].data(), b.p, sizeof(float) * t->dbg[k].size(), hipMemcpyDeviceToHost); };
        t->dbg[0].assign(out.dur.begin(), out.dur.end());
        grab(1, out.f0); grab(2, out.n); grab(3, out.dec); grab(4, out.post); grab(5, out.bert); grab(6, out.d_en); grab(7, out.t_en); grab(8, out.har);
    }
    skw_tts_audio* a = (skw_tts_audio*)malloc(sizeof(skw_tts_audio));
    if (!a) { free(host); return fail("out of memory"); }
    a->samples = host; a->n = (int32_t)n_out; a->sample_rate = SAMPLE_RATE;
    return a;
}
extern "C" const skw_tts_audio* skw_tts_generate(skw_tts* t, const char* text, int32_t sid, float speed) {
    try { return generate_impl(t, text, sid, speed, nullptr, 0); } catch (const std::exception& e) { snprintf(t->errbuf, sizeof t->errbuf, "skw_tts_generate: %s", e.what()); return nullptr; }
}
// the same from token ids (tests and benchmarks: a given number of tokens whatever the lexicon)
extern "C" const skw_tts_audio* skw_tts_generate_ids(skw_tts* t, const int32_t* ids, int32_t n_ids, int32_t sid, float speed) {
    try { return generate_impl(t, nullptr, sid, speed, ids, n_ids); } catch (const std::exception& e) { snprintf(t->errbuf, sizeof t->errbuf, "skw_tts_generate_ids: %s", e.what()); return nullptr; }
}
extern "C" void skw_tts_destroy_audio(const skw_tts_audio* a) { if (!a) return; free((void*)a->samples); free((void*)a); }
