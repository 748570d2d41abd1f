/*
 * oracle/skw_kokoro_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU backend of the Kokoro network (include/skw_kokoro_net.h holds the wiring and the operator contract; this file implements every operator as a plain
 * loop) — the checker of streamkit_amd/csrc/skw_tts.hip, whose operators are HIP kernels.  PARITY UNPINNED: Kokoro-82M's graph and weights are not in
 * /root/reference (the reference calls sherpa-onnx: plugins/native/kokoro/src/ffi.rs:119-137, kokoro_node.rs:581-588) and no reference test holds an audio
 * vector; the architecture is the published one as recalled (header of skw_kokoro_net.h).  Only tests/ may load this.
 */
#include "../include/skw_kokoro_net.h"
#include "../include/skw_math.h"
#include <algorithm>
#include <cstring>
#include <memory>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
using namespace skw::kokoro;

const double TWC[N_FFT] = SKW_KOKORO_TW_COS, TWS[N_FFT] = SKW_KOKORO_TW_SIN;
inline float sigmoid_e(float v) { return 1.0f / (1.0f + skw_expf(-v)); }
inline float tanh_e(float v) { const float e = skw_expf(2.0f * v); return 1.0f - 2.0f / (e + 1.0f); }

struct CpuBackend {
    struct Buf { int T = 0, C = 0; std::shared_ptr<std::vector<float>> p; float* data() const { return p->data(); } };
    static Buf make(int T, int C) { Buf b; b.T = T; b.C = C; b.p = std::make_shared<std::vector<float>>((size_t)T * C, 0.0f); return b; }
    Buf copy(const Buf& x) { Buf b = make(x.T, x.C); memcpy(b.data(), x.data(), sizeof(float) * (size_t)x.T * x.C); return b; }
    Buf embed(const Tensor& tab, const int* ids, int T) {
        const int D = (int)tab.dims[1]; Buf b = make(T, D);
        for (int t = 0; t < T; ++t) memcpy(b.data() + (size_t)t * D, tab.host.data() + (size_t)ids[t] * D, sizeof(float) * D);
        return b;
    }
    Buf add_pos_type(const Buf& x, const Tensor& pos, const Tensor& type) {
        Buf b = make(x.T, x.C);
        for (int t = 0; t < x.T; ++t) for (int c = 0; c < x.C; ++c) b.data()[(size_t)t * x.C + c] = (x.data()[(size_t)t * x.C + c] + pos.host[(size_t)t * x.C + c]) + type.host[c];
        return b;
    }
    Buf conv(const Buf& x, const Tensor& w, const Tensor* bias, int K, int stride, int dil, int pad) {
        const int Co = (int)w.dims[0], Ci = (int)w.dims[1], T = x.T, To = (T + 2 * pad - dil * (K - 1) - 1) / stride + 1;
        Buf o = make(To, Co);
#pragma omp parallel for schedule(static)
        for (int t = 0; t < To; ++t)
            for (int co = 0; co < Co; ++co) {
                float acc = 0.0f;
                for (int tap = 0; tap < K; ++tap) {
                    const int tt = t * stride + tap * dil - pad; if (tt < 0 || tt >= T) continue;
                    const float* xr = x.data() + (size_t)tt * Ci; const float* wr = w.host.data() + ((size_t)co * Ci) * K + tap;
                    for (int ci = 0; ci < Ci; ++ci) acc = fmaf(wr[(size_t)ci * K], xr[ci], acc);
                }
                o.data()[(size_t)t * Co + co] = bias ? acc + bias->host[co] : acc;
            }
        return o;
    }
    Buf convtr(const Buf& x, const Tensor& w, const Tensor* bias, int K, int stride, int pad, int out_pad, bool depthwise) {
        const int Ci = (int)w.dims[0], Co = depthwise ? Ci : (int)w.dims[1], T = x.T, To = (T - 1) * stride - 2 * pad + K + out_pad;
        Buf o = make(To, Co);
#pragma omp parallel for schedule(static)
        for (int u = 0; u < To; ++u)
            for (int co = 0; co < Co; ++co) {
                float acc = 0.0f;
                for (int tap = 0; tap < K; ++tap) {
                    const int num = u + pad - tap; if (num < 0 || num % stride) continue; const int t = num / stride; if (t >= T) continue;
                    const float* xr = x.data() + (size_t)t * Ci;
                    if (depthwise) acc = fmaf(w.host[(size_t)co * K + tap], xr[co], acc);
                    else for (int ci = 0; ci < Ci; ++ci) acc = fmaf(w.host[((size_t)ci * Co + co) * K + tap], xr[ci], acc);
                }
                o.data()[(size_t)u * Co + co] = bias ? acc + bias->host[co] : acc;
            }
        return o;
    }
    /* the f64 summation orders of the contract (skw_kokoro_net.h "statistics"): sum256 = 256 lanes, lane l takes elements l, l + 256, ... ascending, lanes combine by a
     * butterfly within groups of 64 (xor 32, 16, 8, 4, 2, 1) and the four group sums add ascending; chunk512 = runs of 512 elements sequentially, run sums ascending */
    template <class F> static double sum256(int n, F f) {
        double lane[256];
        for (int l = 0; l < 256; ++l) { double a = 0.0; for (int i = l; i < n; i += 256) a += f(i); lane[l] = a; }
        for (int o = 32; o > 0; o >>= 1) { double nx[256]; for (int l = 0; l < 256; ++l) nx[l] = lane[l] + lane[l ^ o]; memcpy(lane, nx, sizeof lane); }
        return lane[0] + lane[64] + lane[128] + lane[192];
    }
    template <class F> static double chunk512(int n, F f) {
        double tot = 0.0;
        for (int i0 = 0; i0 < n; i0 += 512) { double a = 0.0; for (int i = i0; i < std::min(n, i0 + 512); ++i) a += f(i); tot += a; }
        return tot;
    }
    static void row_stats(const float* v, int n, long stride, float* mu, float* rstd, float eps) {
        const bool over_time = stride != 1;      /* instance norm walks time (chunk512), LayerNorm walks channels (sum256) */
        auto val = [&](int i) { return (double)v[(size_t)i * stride]; };
        const double mean = (over_time ? chunk512(n, val) : sum256(n, val)) / n;
        auto sq = [&](int i) { const double u = (double)v[(size_t)i * stride] - mean; return u * u; };
        const double q = over_time ? chunk512(n, sq) : sum256(n, sq);
        *mu = (float)mean; *rstd = (float)(1.0 / sqrt(q / n + (double)eps));
    }
    void layernorm(Buf& x, const Tensor& g, const Tensor& b, float eps) {
        for (int t = 0; t < x.T; ++t) { float* r = x.data() + (size_t)t * x.C; float mu, rs; row_stats(r, x.C, 1, &mu, &rs, eps);
            for (int c = 0; c < x.C; ++c) r[c] = ((r[c] - mu) * rs) * g.host[c] + b.host[c]; }
    }
    Buf style_fc(const Tensor& w, const Tensor& b, const float* s) {
        const int R = (int)w.dims[0]; Buf o = make(1, R);
        for (int r = 0; r < R; ++r) { float acc = 0.0f; for (int j = 0; j < STYLE_DIM; ++j) acc = fmaf(w.host[(size_t)r * STYLE_DIM + j], s[j], acc); o.data()[r] = acc + b.host[r]; }
        return o;
    }
    void ada_ln(Buf& x, const Buf& gb) {
        for (int t = 0; t < x.T; ++t) { float* r = x.data() + (size_t)t * x.C; float mu, rs; row_stats(r, x.C, 1, &mu, &rs, 1e-5f);
            for (int c = 0; c < x.C; ++c) r[c] = ((r[c] - mu) * rs) * (1.0f + gb.data()[c]) + gb.data()[x.C + c]; }
    }
    void ada_in(Buf& x, const Buf& gb) {
#pragma omp parallel for schedule(static)
        for (int c = 0; c < x.C; ++c) { float mu, rs; row_stats(x.data() + c, x.T, x.C, &mu, &rs, 1e-5f);
            for (int t = 0; t < x.T; ++t) { float& v = x.data()[(size_t)t * x.C + c]; v = ((v - mu) * rs) * (1.0f + gb.data()[c]) + gb.data()[x.C + c]; } }
    }
    void ada_in_act(Buf& x, const Buf& gb, Act a, const Tensor* alpha) { ada_in(x, gb); act(x, a, alpha); }
    void add_scale(Buf& a, const Buf& b, float f) { add(a, b); scale(a, f); }
    void act(Buf& x, Act a, const Tensor* alpha) {
        const size_t n = (size_t)x.T * x.C;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) {
            float v = x.data()[i];
            if (a == ACT_LEAKY02) v = v > 0.0f ? v : v * 0.2f;
            else if (a == ACT_LEAKY01) v = v > 0.0f ? v : v * 0.1f;
            else if (a == ACT_LEAKY001) v = v > 0.0f ? v : v * 0.01f;
            else if (a == ACT_GELU) { const float u = 0.79788456080286535588f * (v + 0.044715f * ((v * v) * v)); v = (0.5f * v) * (1.0f + tanh_e(u)); }
            else { const float al = alpha->host[i % x.C]; const float s = sinf(al * v); v = v + (s * s) / al; }
            x.data()[i] = v;
        }
    }
    Buf concat(const std::vector<Buf>& parts) {
        int C = 0; for (auto& p : parts) C += p.C; Buf o = make(parts[0].T, C);
        for (int t = 0; t < o.T; ++t) { int c0 = 0; for (auto& p : parts) { memcpy(o.data() + (size_t)t * C + c0, p.data() + (size_t)t * p.C, sizeof(float) * p.C); c0 += p.C; } }
        return o;
    }
    Buf concat_style(const Buf& x, const float* s) {
        Buf o = make(x.T, x.C + STYLE_DIM);
        for (int t = 0; t < x.T; ++t) { memcpy(o.data() + (size_t)t * o.C, x.data() + (size_t)t * x.C, sizeof(float) * x.C); memcpy(o.data() + (size_t)t * o.C + x.C, s, sizeof(float) * STYLE_DIM); }
        return o;
    }
    void add(Buf& a, const Buf& b) { const size_t n = (size_t)a.T * a.C; for (size_t i = 0; i < n; ++i) a.data()[i] = a.data()[i] + b.data()[i]; }
    void scale(Buf& a, float f) { const size_t n = (size_t)a.T * a.C; for (size_t i = 0; i < n; ++i) a.data()[i] = a.data()[i] * f; }
    Buf upsample2(const Buf& x) { Buf o = make(2 * x.T, x.C); for (int t = 0; t < o.T; ++t) memcpy(o.data() + (size_t)t * x.C, x.data() + (size_t)(t / 2) * x.C, sizeof(float) * x.C); return o; }
    Buf reflect_pad_left(const Buf& x) { Buf o = make(x.T + 1, x.C); memcpy(o.data(), x.data() + (size_t)(x.T > 1 ? 1 : 0) * x.C, sizeof(float) * x.C);
    memcpy(o.data() + x.C, x.data(), sizeof(float) * (size_t)x.T * x.C); return o; }
    Buf attention(const Buf& q, const Buf& k, const Buf& v, int heads) {
        const int T = q.T, C = q.C; Buf o = make(T, C);
#pragma omp parallel for schedule(static) collapse(2)
        for (int h = 0; h < heads; ++h)
            for (int i = 0; i < T; ++i) {
                std::vector<float> s(T); float m = -INFINITY;
                for (int j = 0; j < T; ++j) { float acc = 0.0f; for (int c = 0; c < 64; ++c) acc = fmaf(q.data()[(size_t)i * C + 64 * h + c], k.data()[(size_t)j * C + 64 * h + c], acc);
                s[j] = acc * 0.125f; m = fmaxf(m, s[j]); }
                for (int j = 0; j < T; ++j) s[j] = skw_expf(s[j] - m);
                const double sum = sum256(T, [&](int j) { return (double)s[j]; });
                const float fs = (float)sum; for (int j = 0; j < T; ++j) s[j] = s[j] / fs;
                for (int c = 0; c < 64; ++c) { float acc = 0.0f; for (int j = 0; j < T; ++j) acc = fmaf(s[j], v.data()[(size_t)j * C + 64 * h + c], acc); o.data()[(size_t)i * C + 64 * h + c] = acc; }
            }
        return o;
    }
    Buf lstm_bi(const Buf& x, const Tensor* const* ws) {
        const int H = (int)ws[1]->dims[1], T = x.T; Buf o = make(T, 2 * H);
        for (int dir = 0; dir < 2; ++dir) {
            const Tensor &wih = *ws[4 * dir], &whh = *ws[4 * dir + 1], &bih = *ws[4 * dir + 2], &bhh = *ws[4 * dir + 3];
            Tensor w1; w1.dims = {wih.dims[0], wih.dims[1], 1}; w1.host = wih.host;
            Buf xp = conv(x, w1, &bih, 1, 1, 1, 0);
            std::vector<float> h(H, 0.0f), c(H, 0.0f), a(4 * H);
            for (int s = 0; s < T; ++s) {
                const int t = dir ? T - 1 - s : s;
#pragma omp parallel for schedule(static)
                for (int gi = 0; gi < 4 * H; ++gi) { float acc = 0.0f; for (int kk = 0; kk < H; ++kk) acc = fmaf(whh.host[(size_t)gi * H + kk], h[kk], acc);
                a[gi] = (xp.data()[(size_t)t * 4 * H + gi] + acc) + bhh.host[gi]; }
                for (int j = 0; j < H; ++j) {
                    const float ig = sigmoid_e(a[j]), fg = sigmoid_e(a[H + j]), gg = tanh_e(a[2 * H + j]), og = sigmoid_e(a[3 * H + j]);
                    c[j] = (fg * c[j]) + (ig * gg); h[j] = og * tanh_e(c[j]);
                    o.data()[(size_t)t * 2 * H + dir * H + j] = h[j];
                }
            }
        }
        return o;
    }
    Buf gather_rows(const Buf& x, const std::vector<int>& rows) {
        Buf o = make((int)rows.size(), x.C);
        for (size_t f = 0; f < rows.size(); ++f) memcpy(o.data() + f * x.C, x.data() + (size_t)rows[f] * x.C, sizeof(float) * x.C);
        return o;
    }
    std::vector<int> durations(const Buf& lg, float scale) {
        std::vector<int> d(lg.T);
        for (int t = 0; t < lg.T; ++t) {
            double s = 0.0;
            for (int k = 0; k < lg.C; ++k) s += (double)sigmoid_e(lg.data()[(size_t)t * lg.C + k]);
            const float r = rintf((float)s * scale);
            d[t] = r < 1.0f ? 1 : (int)r;
        }
        return d;
    }
    Buf source_stft(const Buf& f0c, const Tensor& lw, const Tensor& lb) {
        const int M = f0c.T; const long L = (long)M * SRC_UP; const int P = (int)(L / HOP) + 1;
        /* the published phase law (skw_kokoro_net.h header): per harmonic the frame-rate cumulative phase C (inclusive, mod 1), linearly interpolated to the sample rate */
        std::vector<double> C((size_t)N_HARM * M);
        for (int h = 1; h <= N_HARM; ++h) { double c = 0.0; for (int m = 0; m < M; ++m) { double r = (double)h * (double)f0c.data()[m] / (double)SAMPLE_RATE; r -= floor(r); c += r; c -= floor(c);
            C[(size_t)(h - 1) * M + m] = c; } }
        std::vector<float> src(L);
#pragma omp parallel for schedule(static)
        for (long n = 0; n < L; ++n) {
            const int m = (int)(n / SRC_UP); const float f0 = f0c.data()[m];
            const float uv = f0 > 10.0f ? 1.0f : 0.0f; const float amp = f0 > 10.0f ? 0.003f : 0.1f / 3.0f;
            double x = ((double)n + 0.5) / (double)SRC_UP - 0.5; if (x < 0.0) x = 0.0;
            const int m0 = (int)x, m1 = m0 + 1 < M ? m0 + 1 : M - 1; const double wq = x - (double)m0;
            float acc = 0.0f;
            for (int h = 1; h <= N_HARM; ++h) {
                double r1 = 0.0; if (m1 > m0) { r1 = (double)h * (double)f0c.data()[m1] / (double)SAMPLE_RATE; r1 -= floor(r1); }
                double cyc = (double)SRC_UP * (C[(size_t)(h - 1) * M + m0] + wq * r1); cyc -= floor(cyc);
                const float sine = (float)sin(6.283185307179586476925286766559 * cyc) * 0.1f;
                const float val = sine * uv + amp * unit_noise((uint64_t)n * 16 + (uint64_t)h);
                acc = fmaf(lw.host[h - 1], val, acc);
            }
            src[n] = tanh_e(acc + lb.host[0]);
        }
        Buf o = make(P, 2 * N_BINS);
#pragma omp parallel for schedule(static)
        for (int p = 0; p < P; ++p)
            for (int k = 0; k < N_BINS; ++k) {
                double re = 0.0, im = 0.0;
                for (int mm = 0; mm < N_FFT; ++mm) {
                    long idx = (long)p * HOP + mm - N_FFT / 2; if (idx < 0) idx = -idx; if (idx >= L) idx = 2 * (L - 1) - idx;
                    const double wv = (0.5 - 0.5 * TWC[mm]) * (double)src[idx]; const int j = (k * mm) % N_FFT;
                    re += wv * TWC[j]; im -= wv * TWS[j];
                }
                o.data()[(size_t)p * 2 * N_BINS + k] = (float)sqrt(re * re + im * im); o.data()[(size_t)p * 2 * N_BINS + N_BINS + k] = (float)atan2(im, re);
            }
        return o;
    }
    Buf istft(const Buf& post) {
        const int P = post.T; const long n_out = (long)(P - 1) * HOP; Buf y = make((int)n_out, 1);
#pragma omp parallel for schedule(static)
        for (long n = 0; n < n_out; ++n) {
            const long pos = n + N_FFT / 2; double acc = 0.0, wsum = 0.0;
            long p_lo = (pos - (N_FFT - 1) + HOP - 1) / HOP; if (pos - (N_FFT - 1) < 0) p_lo = 0; const long p_hi = pos / HOP;
            for (long p = p_lo; p <= p_hi && p < P; ++p) {
                const int mm = (int)(pos - p * HOP); const double wnd = 0.5 - 0.5 * TWC[mm];
                const float* op = post.data() + (size_t)p * 2 * N_BINS; double xs = 0.0;
                for (int k = 0; k < N_BINS; ++k) {
                    const float mag = skw_expf(op[k]); const float ph = sinf(op[N_BINS + k]);
                    const double re = (double)mag * cos((double)ph), im = (double)mag * sin((double)ph); const int j = (k * mm) % N_FFT;
                    xs += (k == 0) ? re : (k == N_BINS - 1) ? re * TWC[j] : 2.0 * (re * TWC[j] - im * TWS[j]);
                }
                acc += wnd * xs / N_FFT; wsum += wnd * wnd;
            }
            y.data()[n] = wsum > 1e-11 ? (float)(acc / wsum) : 0.0f;
        }
        return y;
    }
};
}  // namespace

extern "C" {
typedef struct { const char* name; const float* data; int32_t n_dims; int64_t dims[4]; } skwo_tts_tensor;
/* taps: every pointer may be NULL; caps in floats.  Returns the number of audio samples (600 x frames), < 0 on error (message in err). */
long skwo_kokoro_forward(const skwo_tts_tensor* tensors, int n_tensors, const int32_t* ids, int T, const float* style, float scale, int max_frames,
                         int32_t* dur_out, int32_t* frames_out, float* bert_out, float* d_en_out, float* t_en_out, float* f0_out, float* n_out, float* dec_out, long dec_cap,
                         float* post_out, long post_cap, float* audio_out, long audio_cap, float* har_out, long har_cap, char* err, int errlen) {
    Weights w;
    for (int i = 0; i < n_tensors; ++i) { Tensor t; long n = 1; for (int k = 0; k < tensors[i].n_dims; ++k) { t.dims.push_back(tensors[i].dims[k]);
    n *= tensors[i].dims[k]; } t.host.assign(tensors[i].data, tensors[i].data + n); w[tensors[i].name] = std::move(t); }
    Dims g; std::string e;
    auto fail = [&](const std::string& m) { if (err && errlen > 0) snprintf(err, errlen, "%s", m.c_str()); return -1L; };
    if (!check(w, &g, &e)) return fail(e);
    CpuBackend be; Net<CpuBackend> net(be, w, g); Outputs<CpuBackend::Buf> out;
    std::vector<int> idv(ids, ids + T);
    if (!net.forward(idv, style, scale, max_frames, &out, &e)) return fail(e);
    auto put = [](float* dst, long cap, const CpuBackend::Buf& b) { if (dst) memcpy(dst, b.data(), sizeof(float) * (size_t)std::min<long>(cap, (long)b.T * b.C)); };
    if (dur_out) for (int t = 0; t < T; ++t) dur_out[t] = out.dur[t];
    if (frames_out) *frames_out = out.F;
    put(bert_out, (long)T * g.hid, out.bert); put(d_en_out, (long)T * g.d, out.d_en); put(t_en_out, (long)T * g.d, out.t_en);
    put(f0_out, 2L * out.F, out.f0); put(n_out, 2L * out.F, out.n); put(dec_out, dec_cap, out.dec); put(post_out, post_cap, out.post);
    put(audio_out, audio_cap, out.audio); put(har_out, har_cap, out.har);
    return (long)out.audio.T;
}
}
