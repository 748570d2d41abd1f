// skw_tts.hip — libskw_tts.so: the speech synthesiser behind the Kokoro TTS node (include/skw_tts.h), hand-written HIP for gfx950.
//
// Replaces the sherpa-onnx calls of /root/reference/plugins/native/kokoro/src/ffi.rs:119-137 (create / generate / destroy).
// PARITY UNPINNED (header of include/skw_tts.h): Kokoro-82M's graph and weights are not in /root/reference; this is a reduced network of the
// same shape — the stages and their arithmetic are specified in DESIGN.md section 7 and restated on the CPU by oracle/skw_kokoro_oracle.c:
//   tokens -> embedding -> n_te x [conv1d k5 -> LayerNorm -> LeakyReLU 0.2]                                  (text encoder)
//   -> AdaLN by the prosody half of the speaker style -> duration = max(1, rint(sum_k sigmoid(proj_k) * length_scale / speed))
//   -> length regulation -> F0 (60 .. 400 Hz) and energy per frame                                             (prosody predictor)
//   -> conv1d k3 over [text features, F0, energy] -> AdaIN(acoustic style) -> n_dec residual AdaIN blocks      (decoder)
//   -> transposed-conv upsampling x120 + harmonic source (<= 8 sines of the running F0 phase) -> snake ResBlock -> conv_post k7
//   -> magnitude = exp, phase = sin -> inverse STFT n_fft 20 / hop 5 / Hann, overlap-add                        (ISTFTNet head, 600 samples per frame at 24 kHz)
// Every contraction accumulates in f64 and rounds once to f32 (the rule include/skw_math.h uses for ggml_norm): the summation order a GPU
// reduction picks then changes nothing a CPU restatement can see, so durations — integers — agree exactly and the waveform to the last few ulps of
// sinf / expf.  These kernels are latency-sized (a sentence is ~100 tokens, ~300 frames); nothing here is on the benchmark's timed path.
#include "../../include/skw_tts.h"
#include "../../include/skw_math.h"
#include "skw_silero.h"      // the ONNX initializer reader (skw::onnx)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define TTS_STYLE 128          // each half of a 256-float style row
#define TTS_U 120              // generator up-sampling: sub-frames per frame
#define TTS_NFFT 20
#define TTS_HOP 5
#define TTS_BINS 11
#define TTS_H 8                // harmonics of the source
#define TTS_RATE 24000
#define TTS_SUBRATE (TTS_RATE / TTS_HOP)      // 4800 sub-frames per second
#define TTS_MAX_TOKENS 510
#define TTS_MAX_FRAMES 6000    // 150 s of audio per call

static void set_err(char* err, size_t n, const char* fmt, ...) { if (!err || !n) return; va_list ap; va_start(ap, fmt); vsnprintf(err, n, fmt, ap); va_end(ap); }

// ------------------------------------------------------------------ kernels (activations are [time][channel] f32 rows)
__global__ void k_tts_embed(const float* emb, const int* ids, int d, float* x) { const int t = blockIdx.x;
for (int c = threadIdx.x; c < d; c += blockDim.x) x[(long)t * d + c] = emb[(long)ids[t] * d + c]; }

// out[t][co] = bias[co] + sum_k sum_ci w[k][ci][co] * in[t + k - K/2][ci]   (zero padding; weight pre-transposed on the host so a wave reads it coalesced)
// pre: 0 none, 1 LeakyReLU(slope) on the input as it is read.  f64 accumulation, ascending (k, ci); one thread per (t, co).
__global__ __launch_bounds__(256) void k_tts_conv1d(const float* in, int T, int Cin, const float* w, const float* bias, int K, int Cout, float* out, int pre, float slope) {
    const int t = blockIdx.x, pad = K / 2;
    extern __shared__ float sh_in[];                           // [K][Cin] window of the input
    for (int i = threadIdx.x; i < K * Cin; i += blockDim.x) { const int k = i / Cin, ci = i % Cin, tt = t + k - pad;
    float v = (tt >= 0 && tt < T) ? in[(long)tt * Cin + ci] : 0.0f; if (pre == 1) v = v > 0.0f ? v : v * slope; sh_in[i] = v; }
    __syncthreads();
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i < K * Cin; ++i) acc += (double)w[(long)i * Cout + co] * (double)sh_in[i];
        out[(long)t * Cout + co] = (float)(acc + (double)(bias ? bias[co] : 0.0f));
    }
}
__device__ __forceinline__ double tts_block_sum(double v, double* sh) {      // blockDim.x == 256
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads(); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v; __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
// LayerNorm over the channels of row t (eps 1e-5, f64 statistics), then: mode 0 gamma/beta + LeakyReLU(0.2); mode 1 AdaLN: * (1 + ada[c]) + ada[d + c]
__global__ __launch_bounds__(256) void k_tts_ln(const float* x, int d, const float* gamma, const float* beta, const float* ada, int mode, float* y) {
    __shared__ double sh[4]; const int t = blockIdx.x; const float* xr = x + (long)t * d;
    double s = 0.0; for (int c = threadIdx.x; c < d; c += 256) s += (double)xr[c];
    const double mean = tts_block_sum(s, sh) / d;
    double q = 0.0; for (int c = threadIdx.x; c < d; c += 256) { const double u = (double)xr[c] - mean; q += u * u; }
    const double var = tts_block_sum(q, sh) / d; const float rstd = (float)(1.0 / sqrt(var + 1e-5)); const float mu = (float)mean;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float n = (xr[c] - mu) * rstd; float v;
        if (mode == 0) { v = n * gamma[c] + beta[c]; v = v > 0.0f ? v : v * 0.2f; } else v = n * (1.0f + ada[c]) + ada[d + c];
        y[(long)t * d + c] = v;
    }
}
// y[r] = bias[r] + sum_j w[r][j] * s[j]   (style projections: 128 inputs)
__global__ void k_tts_style_fc(const float* w, const float* bias, const float* s, int rows, float* y) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= rows) return;
    double acc = 0.0; for (int j = 0; j < TTS_STYLE; ++j) acc += (double)w[(long)r * TTS_STYLE + j] * (double)s[j];
    y[r] = (float)(acc + (double)bias[r]);
}
__device__ __forceinline__ float tts_sigmoid(float v) { return 1.0f / (1.0f + skw_expf(-v)); }     // skw_expf: bit-identical on host and device (include/skw_math.h)
// one block per token: dsum = sum_k sigmoid(dot(w[k], h[t]) + b[k]); dur = max(1, rint(dsum * scale))
__global__ __launch_bounds__(256) void k_tts_duration(const float* h, int d, const float* w, const float* b, int K, float scale, int* dur, float* dsum_out) {
    __shared__ double sh[4]; const int t = blockIdx.x; const float* hr = h + (long)t * d; double tot = 0.0;
    for (int k = 0; k < K; ++k) {
        double s = 0.0; for (int c = threadIdx.x; c < d; c += 256) s += (double)w[(long)k * d + c] * (double)hr[c];
        s = tts_block_sum(s, sh);
        tot += (double)tts_sigmoid((float)(s + (double)b[k]));
    }
    if (threadIdx.x == 0) { const float ds = (float)tot; dsum_out[t] = ds; const float r = rintf(ds * scale); dur[t] = r < 1.0f ? 1 : (int)r; }
}
// per frame: f0 = 60 + 340 sigmoid(w_f0 . h[tok] + v_f0 . s_pr + b), energy = w_n . h[tok] + b_n
__global__ __launch_bounds__(256) void k_tts_f0n(const float* h, int d, const int* tok, const float* wf, const float* vf, const float* bf, const float* wn,
    const float* bn, const float* s_pr, float* f0, float* en) {
    __shared__ double sh[4]; const int f = blockIdx.x; const float* hr = h + (long)tok[f] * d;
    double a = 0.0, e = 0.0; for (int c = threadIdx.x; c < d; c += 256) { a += (double)wf[c] * (double)hr[c]; e += (double)wn[c] * (double)hr[c]; }
    double sv = 0.0; for (int j = threadIdx.x; j < TTS_STYLE; j += 256) sv += (double)vf[j] * (double)s_pr[j];
    a = tts_block_sum(a, sh); e = tts_block_sum(e, sh); sv = tts_block_sum(sv, sh);
    if (threadIdx.x == 0) { f0[f] = 60.0f + 340.0f * tts_sigmoid((float)(a + sv + (double)bf[0])); en[f] = (float)(e + (double)bn[0]); }
}
// decoder input row f = [x[tok(f)][0..d), f0 / 400, energy]
__global__ void k_tts_dec_in(const float* x, int d, const int* tok, const float* f0, const float* en, float* u) {
    const int f = blockIdx.x; const float* xr = x + (long)tok[f] * d; float* ur = u + (long)f * (d + 2);
    for (int c = threadIdx.x; c < d; c += blockDim.x) ur[c] = xr[c];
    if (threadIdx.x == 0) { ur[d] = f0[f] / 400.0f; ur[d + 1] = en[f]; }
}
// instance-norm statistics of channel c over the F frames (biased variance, f64): stats[c] = mean, stats[C + c] = 1 / sqrt(var + 1e-5)
__global__ __launch_bounds__(256) void k_tts_inorm_stats(const float* z, int F, int C, float* stats) {
    __shared__ double sh[4]; const int c = blockIdx.x;
    double s = 0.0; for (int f = threadIdx.x; f < F; f += 256) s += (double)z[(long)f * C + c];
    const double mean = tts_block_sum(s, sh) / F;
    double q = 0.0; for (int f = threadIdx.x; f < F; f += 256) { const double u = (double)z[(long)f * C + c] - mean; q += u * u; }
    const double var = tts_block_sum(q, sh) / F;
    if (threadIdx.x == 0) { stats[c] = (float)mean; stats[C + c] = (float)(1.0 / sqrt(var + 1e-5)); }
}
// AdaIN + LeakyReLU(0.2): r = leaky(((z - mean) * rstd) * (1 + ada[c]) + ada[C + c]); out = res ? (res + r) * rsqrt(2) : r
__global__ void k_tts_adain(const float* z, long n, int C, const float* stats, const float* ada, const float* res, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const int c = (int)(i % C);
    float v = ((z[i] - stats[c]) * stats[C + c]) * (1.0f + ada[c]) + ada[C + c]; v = v > 0.0f ? v : v * 0.2f;
    out[i] = res ? (res[i] + v) * 0.70710678118654752f : v;
}
// running phase of the source in cycles at the start of every frame: Phi[f] = frac(sum_{j<f} U * f0[j] / 4800), f64, one lane (F <= 6000)
__global__ void k_tts_phase_scan(const float* f0, int F, double* phi) { double a = 0.0;
for (int f = 0; f < F; ++f) { phi[f] = a; a += (double)TTS_U * (double)f0[f] / (double)TTS_SUBRATE; a -= floor(a); } }
// generator input at sub-frame p = f * U + u: g[p][cg] = b[cg] + sum_c wup[u][c][cg] * z[f][c] + sum_h wsrc[h][cg] * har_h(p);
// har_h = sin(2 pi frac((h + 1) * (Phi[f] + u * f0[f] / 4800))) for (h + 1) * f0[f] < 2400 Hz, else 0   (phase in f64)
__global__ __launch_bounds__(64) void k_tts_gen_in(const float* z, int C, int G, const float* wup, const float* bup, const float* wsrc, const float* f0, const double* phi, float* g) {
    const int p = blockIdx.x, f = p / TTS_U, u = p % TTS_U; __shared__ float har[TTS_H];
    if (threadIdx.x < TTS_H) {
        const int h = threadIdx.x; const double ph = phi[f] + (double)u * (double)f0[f] / (double)TTS_SUBRATE; double cyc = (double)(h + 1) * ph; cyc -= floor(cyc);
        har[h] = ((float)(h + 1) * f0[f] < 0.5f * (float)TTS_SUBRATE) ? (float)sin(6.283185307179586476925286766559 * cyc) : 0.0f;
    }
    __syncthreads();
    for (int cg = threadIdx.x; cg < G; cg += 64) {
        double acc = 0.0; const float* zr = z + (long)f * C;
        for (int c = 0; c < C; ++c) acc += (double)wup[((long)u * C + c) * G + cg] * (double)zr[c];
        double hs = 0.0; for (int h = 0; h < TTS_H; ++h) hs += (double)wsrc[h * G + cg] * (double)har[h];
        g[(long)p * G + cg] = (float)(acc + hs + (double)bup[cg]);
    }
}
// snake activation: x + sin^2(alpha_c x) / alpha_c
__global__ void k_tts_snake(const float* g, long n, int G, const float* alpha, float* out) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
if (i >= n) return; const float a = alpha[i % G]; const float s = sinf(a * g[i]); out[i] = g[i] + s * s / a; }
__global__ void k_tts_add(const float* a, const float* b, long n, float* out) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[i] + b[i]; }
// ISTFTNet head: o[p][0..11) -> magnitude exp, o[p][11..22) -> phase sin; inverse real DFT of each sub-frame (N = 20), periodic Hann window,
// overlap-add with hop 5 normalised by the summed squared window, centre-trimmed: sample n (0 <= n < 5 (P - 1)) sits at n + 10 of the untrimmed signal.
__global__ void k_tts_istft(const float* o, int P, float* y, long n_out) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x; if (n >= n_out) return;
    const long pos = n + TTS_NFFT / 2; double acc = 0.0, wsum = 0.0;
    const long p_hi = pos / TTS_HOP, p_lo = (pos - (TTS_NFFT - 1) + TTS_HOP - 1) / TTS_HOP;      // frames p with 0 <= pos - 5 p < 20
    for (long p = (p_lo < 0 ? 0 : p_lo); p <= p_hi && p < P; ++p) {
        const int m = (int)(pos - p * TTS_HOP);
        const double wnd = 0.5 - 0.5 * cos(6.283185307179586476925286766559 * m / TTS_NFFT);
        const float* op = o + p * (2 * TTS_BINS); double x = 0.0;
        for (int k = 0; k < TTS_BINS; ++k) {
            const float mag = skw_expf(op[k]); const float ph = sinf(op[TTS_BINS + k]);
            const double re = (double)mag * cos((double)ph), im = (double)mag * sin((double)ph);
            const double ang = 6.283185307179586476925286766559 * k * m / TTS_NFFT;
            const double term = re * cos(ang) - im * sin(ang);
            x += (k == 0 || k == TTS_BINS - 1) ? (k == 0 ? re : re * cos(ang)) : 2.0 * term;      // bins 0 and N/2 are real in an inverse real DFT
        }
        acc += wnd * x / TTS_NFFT; wsum += wnd * wnd;
    }
    y[n] = wsum > 1e-11 ? (float)(acc / wsum) : 0.0f;
}

// ------------------------------------------------------------------ host
struct DevT { float* p = nullptr; std::vector<int64_t> dims; };
struct skw_tts {
    int device = 0; hipStream_t stream = nullptr; std::mutex mu; char errbuf[512] = {0};
    std::map<std::string, DevT> w; std::vector<void*> allocs;
    int n_sym = 0, d = 0, n_te = 0, K = 0, C = 0, n_dec = 0, G = 0; float length_scale = 1.0f;
    float* voices = nullptr; int n_spk = 0, voice_rows = 0;
    std::map<unsigned, int> sym2id; std::map<std::string, std::vector<int>> lexicon;       // code point -> id; lower-case word -> ids
    std::vector<float> dbg[5]; float last_ms = 0.0f;
    // scratch (grown on demand)
    std::vector<std::pair<void**, size_t>> scratch;
};
static float* upload(skw_tts* t, const float* h, size_t n) { float* d = nullptr;
if (hipMalloc((void**)&d, std::max<size_t>(1, n) * 4) != hipSuccess) return nullptr;
if (n && hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; } t->allocs.push_back(d); return d; }

static bool read_file(const char* path, std::vector<uint8_t>* out, size_t limit) {
    FILE* f = fopen(path, "rb"); if (!f) return false; uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) { out->insert(out->end(), buf, buf + n); if (out->size() > limit) { fclose(f); return false; } }
    fclose(f); return true;
}
static unsigned next_cp(const std::string& s, size_t* i) {
    const unsigned char* p = (const unsigned char*)s.data(); const unsigned char c = p[*i]; int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
    if (*i + len > s.size()) len = 1; unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1)); for (int k = 1; k < len; ++k) cp = (cp << 6) | (p[*i + k] & 0x3F); *i += len; return cp;
}
// tokens.txt: "<symbol> <id>" per line; a line that starts with a space names the space symbol (sherpa-onnx's convention)
static bool load_tokens(skw_tts* t, const char* path, std::string* err) {
    std::vector<uint8_t> b; if (!read_file(path, &b, 16u << 20)) { *err = std::string("cannot read tokens file ") + path; return false; }
    std::string s((const char*)b.data(), b.size()); size_t i = 0;
    while (i < s.size()) {
        size_t e = s.find('\n', i); if (e == std::string::npos) e = s.size(); std::string line = s.substr(i, e - i); i = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back(); if (line.empty()) continue;
        const size_t sp = line.rfind(' '); if (sp == std::string::npos) continue;
        std::string sym = line.substr(0, sp); const int id = atoi(line.c_str() + sp + 1); if (sym.empty()) sym = " ";
        size_t k = 0; const unsigned cp = next_cp(sym, &k); if (k == sym.size()) t->sym2id[cp] = id;       // single-code-point symbols (all of Kokoro's are)
    }
    if (t->sym2id.empty()) { *err = std::string("no symbols in tokens file ") + path; return false; }
    return true;
}
static std::string lower_ascii(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }
static void load_lexicon(skw_tts* t, const char* list) {      // "word ph ph ..." per line; the phonemes are symbols of tokens.txt
    if (!list) return; std::string all = list; size_t i = 0;
    while (i <= all.size()) {
        size_t e = all.find(',', i); if (e == std::string::npos) e = all.size(); const std::string path = all.substr(i, e - i); i = e + 1; if (path.empty()) continue;
        std::vector<uint8_t> b; if (!read_file(path.c_str(), &b, 256u << 20)) continue;      // missing lexicon files are not an error (kokoro_node.rs never checks them)
        std::string s((const char*)b.data(), b.size()); size_t j = 0;
        while (j < s.size()) {
            size_t le = s.find('\n', j); if (le == std::string::npos) le = s.size(); std::string line = s.substr(j, le - j); j = le + 1;
            const size_t sp = line.find_first_of(" \t"); if (sp == std::string::npos || sp == 0) continue;
            const std::string word = lower_ascii(line.substr(0, sp)); if (t->lexicon.count(word)) continue;       // first entry wins
            std::vector<int> ids; for (size_t k = sp; k < line.size();) { const unsigned cp = next_cp(line, &k);
            if (cp == ' ' || cp == '\t' || cp == '\r') continue; auto it = t->sym2id.find(cp); if (it != t->sym2id.end()) ids.push_back(it->second); }
            if (!ids.empty()) t->lexicon[word] = ids;
        }
    }
}
// text -> ids: words found in the lexicon become their phoneme ids, everything else goes code point by code point through tokens.txt
// (unknown symbols are dropped); pad id 0 at both ends; at most TTS_MAX_TOKENS
static std::vector<int> tokenize(const skw_tts* t, const std::string& text) {
    std::vector<int> ids; ids.push_back(0);
    size_t i = 0;
    while (i < text.size() && (int)ids.size() < TTS_MAX_TOKENS - 1) {
        size_t j = i; std::string word;
        while (j < text.size()) { const unsigned char c = (unsigned char)text[j]; if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '\'') { word.push_back((char)c); ++j; } else break; }
        if (!word.empty()) {
            auto it = t->lexicon.find(lower_ascii(word));
            if (it != t->lexicon.end()) { for (int id : it->second) if ((int)ids.size() < TTS_MAX_TOKENS - 1) ids.push_back(id); i = j; continue; }
        }
        const unsigned cp = next_cp(text, &i);
        auto it = t->sym2id.find(cp); if (it == t->sym2id.end() && cp >= 'A' && cp <= 'Z') it = t->sym2id.find(cp - 'A' + 'a');
        if (it != t->sym2id.end()) ids.push_back(it->second);
    }
    ids.push_back(0); return ids;
}

static bool want(skw_tts* t, const skw::onnx::Model& m, const std::string& name, std::initializer_list<int64_t> dims, std::string* err, bool transpose_conv = false, bool transpose_ups = false) {
    for (const auto& x : m.tensors) if (x.name == name && !x.data.empty()) {
        if (dims.size() && !x.is(dims)) { *err = "tensor '" + name + "' has an unexpected shape"; return false; }
        DevT dt; dt.dims = x.dims;
        if (transpose_conv) {          // [Cout][Cin][K] -> [K][Cin][Cout]
            const int64_t Co = x.dims[0], Ci = x.dims[1], Kk = x.dims[2]; std::vector<float> h(x.data.size());
            for (int64_t co = 0; co < Co; ++co) for (int64_t ci = 0; ci < Ci; ++ci) for (int64_t k = 0; k < Kk; ++k) h[(size_t)((k * Ci + ci) * Co + co)] = x.data[(size_t)((co * Ci + ci) * Kk + k)];
            dt.p = upload(t, h.data(), h.size());
        } else if (transpose_ups) {    // ConvTranspose1d [Cin][Cout][U] -> [U][Cin][Cout]
            const int64_t Ci = x.dims[0], Co = x.dims[1], U = x.dims[2]; std::vector<float> h(x.data.size());
            for (int64_t ci = 0; ci < Ci; ++ci) for (int64_t co = 0; co < Co; ++co) for (int64_t u = 0; u < U; ++u) h[(size_t)((u * Ci + ci) * Co + co)] = x.data[(size_t)((ci * Co + co) * U + u)];
            dt.p = upload(t, h.data(), h.size());
        } else dt.p = upload(t, x.data.data(), x.data.size());
        if (!dt.p) { *err = "device allocation failed for '" + name + "'"; return false; }
        t->w[name] = dt; return true;
    }
    *err = "missing tensor '" + name + "' in the model"; return false;
}
static const skw::onnx::Tensor* find_t(const skw::onnx::Model& m, const std::string& name) { for (const auto& x : m.tensors) if (x.name == name && !x.data.empty()) return &x; return nullptr; }

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
    if (!load_tokens(t, cfg->tokens, &e)) return fail(e);
    load_lexicon(t, cfg->lexicon);
    std::vector<uint8_t> bytes; if (!read_file(cfg->model, &bytes, 1024u << 20)) return fail(std::string("cannot read model file ") + cfg->model);
    skw::onnx::Model m; if (!skw::onnx::parse_model(bytes, &m, &e)) return fail(std::string("model file ") + cfg->model + ": " + e);
    const skw::onnx::Tensor* emb = find_t(m, "text_encoder.embedding.weight");
    if (!emb || emb->dims.size() != 2) return fail("model file: no text_encoder.embedding.weight [n_sym, d] (this build reads its own reduced Kokoro-shaped network, DESIGN.md section 7; "
                                                       "a Kokoro-82M export is not supported yet)");
    t->n_sym = (int)emb->dims[0]; t->d = (int)emb->dims[1];
    if (t->d < 16 || t->d > 1024 || t->n_sym < 2) return fail("model file: implausible embedding shape");
    for (auto& kv : t->sym2id) if (kv.second < 0 || kv.second >= t->n_sym) return fail("tokens file names an id outside the embedding table");
    const int d = t->d; bool ok = want(t, m, "text_encoder.embedding.weight", {t->n_sym, d}, &e);
    while (ok && find_t(m, "text_encoder.cnn." + std::to_string(t->n_te) + ".weight")) {
        const std::string p = "text_encoder.cnn." + std::to_string(t->n_te) + ".";
        ok = want(t, m, p + "weight", {d, d, 5}, &e, true) && want(t, m, p + "bias", {d}, &e) && want(t, m, p + "norm.gamma", {d}, &e) && want(t, m, p + "norm.beta", {d}, &e); t->n_te++;
    }
    const skw::onnx::Tensor* dp = find_t(m, "predictor.duration_proj.weight"); if (ok && (!dp || dp->dims.size() != 2)) { ok = false; e = "missing tensor 'predictor.duration_proj.weight'"; }
    if (ok) t->K = (int)dp->dims[0];
    ok = ok && want(t, m, "predictor.text_encoder.fc.weight", {2 * d, TTS_STYLE}, &e) && want(t, m, "predictor.text_encoder.fc.bias", {2 * d}, &e)
            && want(t, m, "predictor.duration_proj.weight", {t->K, d}, &e) && want(t, m, "predictor.duration_proj.bias", {t->K}, &e)
            && want(t, m, "predictor.F0_proj.weight", {d}, &e) && want(t, m, "predictor.F0_proj.style", {TTS_STYLE}, &e) && want(t, m, "predictor.F0_proj.bias", {1}, &e)
            && want(t, m, "predictor.N_proj.weight", {d}, &e) && want(t, m, "predictor.N_proj.bias", {1}, &e);
    const skw::onnx::Tensor* de = find_t(m, "decoder.encode.weight"); if (ok && (!de || de->dims.size() != 3)) { ok = false; e = "missing tensor 'decoder.encode.weight'"; }
    if (ok) t->C = (int)de->dims[0];
    const int C = t->C;
    ok = ok && want(t, m, "decoder.encode.weight", {C, d + 2, 3}, &e, true) && want(t, m, "decoder.encode.bias", {C}, &e) && want(t, m, "decoder.encode.fc.weight",
        {2 * C, TTS_STYLE}, &e) && want(t, m, "decoder.encode.fc.bias", {2 * C}, &e);
    while (ok && find_t(m, "decoder.decode." + std::to_string(t->n_dec) + ".weight")) {
        const std::string p = "decoder.decode." + std::to_string(t->n_dec) + ".";
        ok = want(t, m, p + "weight", {C, C, 3}, &e, true) && want(t, m, p + "bias", {C}, &e) && want(t, m, p + "fc.weight", {2 * C, TTS_STYLE}, &e) && want(t, m, p + "fc.bias", {2 * C}, &e);
        t->n_dec++;
    }
    const skw::onnx::Tensor* up = find_t(m, "decoder.generator.ups.weight"); if (ok && (!up || up->dims.size() != 3)) { ok = false; e = "missing tensor 'decoder.generator.ups.weight'"; }
    if (ok) t->G = (int)up->dims[1];
    const int G = t->G;
    ok = ok && want(t, m, "decoder.generator.ups.weight", {C, G, TTS_U}, &e, false, true) && want(t, m, "decoder.generator.ups.bias", {G}, &e)
            && want(t, m, "decoder.generator.source.weight", {TTS_H, G}, &e) && want(t, m, "decoder.generator.resblock.alpha", {G}, &e)
            && want(t, m, "decoder.generator.resblock.weight", {G, G, 3}, &e, true) && want(t, m, "decoder.generator.resblock.bias", {G}, &e)
            && want(t, m, "decoder.generator.conv_post.weight", {2 * TTS_BINS, G, 7}, &e, true) && want(t, m, "decoder.generator.conv_post.bias", {2 * TTS_BINS}, &e);
    if (!ok) return fail(std::string("model file ") + cfg->model + ": " + e);
    if (C < 4 || C > 1024 || G < 4 || G > 512 || t->K < 1 || t->K > 256) return fail("model file: implausible layer sizes");
    {   // voices.bin: f32 [n_spk][rows][256]; rows = 510 in Kokoro's files
        std::vector<uint8_t> vb; if (!read_file(cfg->voices, &vb, 1024u << 20)) return fail(std::string("cannot read voices file ") + cfg->voices);
        const size_t row = 2 * TTS_STYLE * 4; if (vb.size() < row || vb.size() % row) return fail("voices file: size is not a multiple of 256 floats");
        const size_t rows = vb.size() / row; t->voice_rows = rows % TTS_MAX_TOKENS == 0 ? TTS_MAX_TOKENS : 1; t->n_spk = (int)(rows / t->voice_rows);
        t->voices = upload(t, (const float*)vb.data(), vb.size() / 4); if (!t->voices) return fail("device allocation failed for the voices");
    }
    hipDeviceSynchronize();
    return t;
}
extern "C" skw_tts* skw_tts_create(const skw_tts_config* cfg, char* err, size_t errlen) {
    try { return create_impl(cfg, err, errlen); } catch (const std::exception& e) { set_err(err, errlen, "skw_tts_create: %s", e.what()); return nullptr; }
}
extern "C" void skw_tts_destroy(skw_tts* t) { if (!t) return; hipSetDevice(t->device); if (t->stream) { hipStreamSynchronize(t->stream);
hipStreamDestroy(t->stream); } for (void* p : t->allocs) hipFree(p); delete t; }
extern "C" const char* skw_tts_last_error(const skw_tts* t) { return t->errbuf; }
extern "C" int32_t skw_tts_num_speakers(const skw_tts* t) { return t->n_spk; }
extern "C" int32_t skw_tts_sample_rate(const skw_tts*) { return TTS_RATE; }
extern "C" float skw_tts_last_ms(const skw_tts* t) { return t->last_ms; }
extern "C" int32_t skw_tts_tokenize(skw_tts* t, const char* text, int32_t* ids, int32_t cap) {
    try { std::vector<int> v = tokenize(t, text ? text : ""); const int n = std::min((int)v.size(), (int)cap); for (int i = 0; i < n; ++i) ids[i] = v[i]; return n; } catch (...) { return -1; }
}
extern "C" long skw_tts_debug_get(skw_tts* t, int what, float* out, long cap) {
    if (what < 0 || what > 4) return -1; std::lock_guard<std::mutex> l(t->mu);
    const auto& v = t->dbg[what]; if (out) memcpy(out, v.data(), sizeof(float) * std::min<long>(cap, (long)v.size())); return (long)v.size();
}

struct Scratch { std::vector<void*> p; ~Scratch() { for (void* q : p) hipFree(q);
} template <typename T> T* get(size_t n) { T* d = nullptr; if (hipMalloc((void**)&d, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) return nullptr;
p.push_back(d); return d; } };
#define TCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(t->errbuf, 512, "HIP error '%s' at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); return nullptr; } } while (0)
#define TNULL(p) do { if (!(p)) { snprintf(t->errbuf, 512, "device allocation failed at %s:%d", __FILE__, __LINE__); return nullptr; } } while (0)

static const skw_tts_audio* generate_impl(skw_tts* t, const char* text, int32_t sid, float speed) {
    std::lock_guard<std::mutex> l(t->mu); t->errbuf[0] = 0;
    if (!text) { snprintf(t->errbuf, 512, "null text"); return nullptr; }
    if (!(speed > 0.0f) || !std::isfinite(speed)) { snprintf(t->errbuf, 512, "speed must be positive"); return nullptr; }
    if (sid < 0 || sid >= t->n_spk) { snprintf(t->errbuf, 512, "speaker id %d outside [0, %d)", sid, t->n_spk); return nullptr; }
    TCHK(hipSetDevice(t->device));
    const std::vector<int> ids = tokenize(t, text); const int T = (int)ids.size();
    if (T <= 2) { snprintf(t->errbuf, 512, "no symbol of the text is in the model's token table"); return nullptr; }
    const int d = t->d, C = t->C, G = t->G, K = t->K; hipStream_t s = t->stream; Scratch sc;
    auto W = [&](const std::string& n) { return t->w[n].p; };
    hipEvent_t e0, e1; TCHK(hipEventCreate(&e0)); TCHK(hipEventCreate(&e1)); TCHK(hipEventRecord(e0, s));
    int* d_ids = sc.get<int>(T); float* x = sc.get<float>((size_t)T * d); float* y = sc.get<float>((size_t)T * d); float* h = sc.get<float>((size_t)T * d);
    float* ada = sc.get<float>(2 * (size_t)std::max(d, C)); int* d_dur = sc.get<int>(T); float* d_dsum = sc.get<float>(T);
    TNULL(d_ids && x && y && h && ada && d_dur && d_dsum);
    TCHK(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * T, hipMemcpyHostToDevice, s));
    const int row = std::min(T - 2, t->voice_rows - 1);      // the style row is chosen by the token count (Kokoro's voices are indexed by length)
    const float* style = t->voices + ((size_t)sid * t->voice_rows + (size_t)std::max(0, row)) * 2 * TTS_STYLE; const float* s_ac = style; const float* s_pr = style + TTS_STYLE;
    hipLaunchKernelGGL(k_tts_embed, dim3(T), dim3(128), 0, s, W("text_encoder.embedding.weight"), d_ids, d, x);
    for (int i = 0; i < t->n_te; ++i) {
        const std::string p = "text_encoder.cnn." + std::to_string(i) + ".";
        hipLaunchKernelGGL(k_tts_conv1d, dim3(T), dim3(256), sizeof(float) * 5 * d, s, x, T, d, W(p + "weight"), W(p + "bias"), 5, d, y, 0, 0.0f);
        hipLaunchKernelGGL(k_tts_ln, dim3(T), dim3(256), 0, s, y, d, W(p + "norm.gamma"), W(p + "norm.beta"), nullptr, 0, x);
    }
    hipLaunchKernelGGL(k_tts_style_fc, dim3((2 * d + 63) / 64), dim3(64), 0, s, W("predictor.text_encoder.fc.weight"), W("predictor.text_encoder.fc.bias"), s_pr, 2 * d, ada);
    hipLaunchKernelGGL(k_tts_ln, dim3(T), dim3(256), 0, s, x, d, nullptr, nullptr, ada, 1, h);
    hipLaunchKernelGGL(k_tts_duration, dim3(T), dim3(256), 0, s, h, d, W("predictor.duration_proj.weight"), W("predictor.duration_proj.bias"), K, t->length_scale / speed, d_dur, d_dsum);
    std::vector<int> dur(T); std::vector<float> dsum(T);
    TCHK(hipMemcpyAsync(dur.data(), d_dur, sizeof(int) * T, hipMemcpyDeviceToHost, s));
    TCHK(hipMemcpyAsync(dsum.data(), d_dsum, sizeof(float) * T, hipMemcpyDeviceToHost, s)); TCHK(hipStreamSynchronize(s));
    // length regulation on the host (T <= 510 integers): frame f belongs to token tok[f]
    std::vector<int> tok; for (int i = 0; i < T; ++i) for (int k = 0; k < dur[i] && (int)tok.size() < TTS_MAX_FRAMES; ++k) tok.push_back(i);
    const int F = (int)tok.size(); const long P = (long)F * TTS_U; const long n_out = TTS_HOP * (P - 1);
    int* d_tok = sc.get<int>(F); float* f0 = sc.get<float>(F); float* en = sc.get<float>(F); float* u = sc.get<float>((size_t)F * (d + 2));
    float* z = sc.get<float>((size_t)F * C); float* r = sc.get<float>((size_t)F * C);
    float* z2 = sc.get<float>((size_t)F * C); float* stats = sc.get<float>(2 * (size_t)C); double* phi = sc.get<double>(F);
    float* g = sc.get<float>((size_t)P * G); float* g1 = sc.get<float>((size_t)P * G);
    float* g2 = sc.get<float>((size_t)P * G); float* o = sc.get<float>((size_t)P * 2 * TTS_BINS); float* yv = sc.get<float>((size_t)n_out);
    TNULL(d_tok && f0 && en && u && z && r && z2 && stats && phi && g && g1 && g2 && o && yv);
    TCHK(hipMemcpyAsync(d_tok, tok.data(), sizeof(int) * F, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_tts_f0n, dim3(F), dim3(256), 0, s, h, d, d_tok, W("predictor.F0_proj.weight"), W("predictor.F0_proj.style"), W("predictor.F0_proj.bias"),
        W("predictor.N_proj.weight"), W("predictor.N_proj.bias"), s_pr, f0, en);
    hipLaunchKernelGGL(k_tts_dec_in, dim3(F), dim3(128), 0, s, x, d, d_tok, f0, en, u);
    hipLaunchKernelGGL(k_tts_conv1d, dim3(F), dim3(256), sizeof(float) * 3 * (d + 2), s, u, F, d + 2, W("decoder.encode.weight"), W("decoder.encode.bias"), 3, C, r, 0, 0.0f);
    const long nz = (long)F * C;
    hipLaunchKernelGGL(k_tts_style_fc, dim3((2 * C + 63) / 64), dim3(64), 0, s, W("decoder.encode.fc.weight"), W("decoder.encode.fc.bias"), s_ac, 2 * C, ada);
    hipLaunchKernelGGL(k_tts_inorm_stats, dim3(C), dim3(256), 0, s, r, F, C, stats);
    hipLaunchKernelGGL(k_tts_adain, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, r, nz, C, stats, ada, nullptr, z);
    for (int i = 0; i < t->n_dec; ++i) {
        const std::string p = "decoder.decode." + std::to_string(i) + ".";
        hipLaunchKernelGGL(k_tts_conv1d, dim3(F), dim3(256), sizeof(float) * 3 * C, s, z, F, C, W(p + "weight"), W(p + "bias"), 3, C, r, 0, 0.0f);
        hipLaunchKernelGGL(k_tts_style_fc, dim3((2 * C + 63) / 64), dim3(64), 0, s, W(p + "fc.weight"), W(p + "fc.bias"), s_ac, 2 * C, ada);
        hipLaunchKernelGGL(k_tts_inorm_stats, dim3(C), dim3(256), 0, s, r, F, C, stats);
        hipLaunchKernelGGL(k_tts_adain, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, r, nz, C, stats, ada, z, z2);
        std::swap(z, z2);
    }
    hipLaunchKernelGGL(k_tts_phase_scan, dim3(1), dim3(1), 0, s, f0, F, phi);
    hipLaunchKernelGGL(k_tts_gen_in, dim3((unsigned)P), dim3(64), 0, s, z, C, G, W("decoder.generator.ups.weight"), W("decoder.generator.ups.bias"), W("decoder.generator.source.weight"), f0, phi, g);
    const long ng = P * G;
    hipLaunchKernelGGL(k_tts_snake, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, g, ng, G, W("decoder.generator.resblock.alpha"), g1);
    hipLaunchKernelGGL(k_tts_conv1d, dim3((unsigned)P), dim3(64), sizeof(float) * 3 * G, s, g1, (int)P, G, W("decoder.generator.resblock.weight"),
        W("decoder.generator.resblock.bias"), 3, G, g2, 0, 0.0f);
    hipLaunchKernelGGL(k_tts_add, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, g, g2, ng, g1);
    hipLaunchKernelGGL(k_tts_conv1d, dim3((unsigned)P), dim3(64), sizeof(float) * 7 * G, s, g1, (int)P, G, W("decoder.generator.conv_post.weight"),
        W("decoder.generator.conv_post.bias"), 7, 2 * TTS_BINS, o, 1, 0.01f);
    hipLaunchKernelGGL(k_tts_istft, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, o, (int)P, yv, n_out);
    TCHK(hipEventRecord(e1, s));
    float* host = (float*)malloc(sizeof(float) * (size_t)std::max<long>(1, n_out)); if (!host) { snprintf(t->errbuf, 512, "out of memory"); return nullptr; }
    if (hipMemcpyAsync(host, yv, sizeof(float) * (size_t)n_out, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { free(host);
    snprintf(t->errbuf, 512, "synthesis kernels failed"); return nullptr; }
    hipEventElapsedTime(&t->last_ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
    {   // stage taps for the parity tests (small: a sentence)
        t->dbg[0].assign(dur.begin(), dur.end());
        t->dbg[1].resize(F); t->dbg[2].resize(F); t->dbg[3].resize((size_t)nz); t->dbg[4].resize((size_t)P * 2 * TTS_BINS);
        hipMemcpy(t->dbg[1].data(), f0, sizeof(float) * F, hipMemcpyDeviceToHost); hipMemcpy(t->dbg[2].data(), en, sizeof(float) * F, hipMemcpyDeviceToHost);
        hipMemcpy(t->dbg[3].data(), z, sizeof(float) * nz, hipMemcpyDeviceToHost); hipMemcpy(t->dbg[4].data(), o, sizeof(float) * (size_t)P * 2 * TTS_BINS, hipMemcpyDeviceToHost);
    }
    skw_tts_audio* a = (skw_tts_audio*)malloc(sizeof(skw_tts_audio)); if (!a) { free(host); return nullptr; }
    a->samples = host; a->n = (int32_t)n_out; a->sample_rate = TTS_RATE; return a;
}
extern "C" const skw_tts_audio* skw_tts_generate(skw_tts* t, const char* text, int32_t sid, float speed) {
    if (!t) return nullptr;
    try { return generate_impl(t, text, sid, speed); } catch (const std::exception& e) { snprintf(t->errbuf, 512, "skw_tts_generate: %s", e.what()); return nullptr; }
}
extern "C" void skw_tts_destroy_audio(const skw_tts_audio* a) { if (!a) return; free((void*)a->samples); free((void*)a); }
