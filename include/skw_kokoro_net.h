/*
 * skw_kokoro_net.h — the Kokoro-82M network as a sequence of operator calls, shared by the product (streamkit_amd/csrc/skw_tts.hip: every
 * operator a HIP kernel) and by its checker (oracle/skw_kokoro_oracle.cpp: every operator a plain loop).  Only the WIRING lives here; the
 * arithmetic of each operator is specified below and implemented twice, once per backend.
 *
 * PARITY UNPINNED.  What the reference runs behind SherpaOnnxOfflineTtsGenerate (/root/reference/plugins/native/kokoro/src/kokoro_node.rs:581-588,
 * ffi.rs:119-137) is Kokoro-82M's ONNX graph inside onnxruntime; neither is in /root/reference or offline.  The architecture restated here is the
 * published one (StyleTTS2 family, hexgrad/Kokoro-82M), RECALLED — module list, tensor shapes and forward order:
 *   bert            ALBERT: embeddings (word + position + token type, 128) -> LayerNorm -> 128 -> 768 -> n_layers x ONE shared layer
 *                   (12 heads x 64 self attention + dense, LayerNorm; FFN 768 -> 2048 gelu_new -> 768, LayerNorm)
 *   bert_encoder    Linear 768 -> 512
 *   predictor       DurationEncoder 3 x [BiLSTM(512 + 128 -> 2 x 256), AdaLayerNorm(style)], duration BiLSTM + Linear 512 -> 50 ("sigmoid().sum()" durations),
 *                   shared BiLSTM over frames, F0 / N: 3 AdainResBlk1d each (512 -> 512, 512 -> 256 with x2 up-sampling, 256 -> 256) + 1x1 projection
 *   text_encoder    Embedding 178 x 512 -> 3 x [Conv1d k5, LayerNorm over channels, LeakyReLU 0.2] -> BiLSTM(512 -> 2 x 256)
 *   decoder         F0_conv / N_conv (k3, stride 2), asr_res 512 -> 64, encode AdainResBlk1d(514 -> 1024), decode 3 x (1090 -> 1024) + (1090 -> 512, x2)
 *   generator       ISTFTNet: harmonic source (9 sines of the up-sampled F0, linear merge, tanh) -> STFT(n_fft 20, hop 5) -> noise_convs + AdaINResBlock1;
 *                   2 x [LeakyReLU 0.1, ConvTranspose1d (x10 k20, x6 k12), + source branch, mean of 3 AdaINResBlock1 (k 3 / 7 / 11, dilations 1 / 3 / 5, Snake)],
 *                   LeakyReLU, conv_post k7 -> 11 log-magnitudes + 11 phases -> inverse STFT: 600 samples per predicted frame at 24 kHz
 * All widths are read from the tensors' shapes, so a reduced model (tools/make_synth_kokoro.py "micro") runs the same code as the 82 M-parameter geometry.
 * Tensor names follow the PyTorch modules (weight-norm pairs folded into `.weight`, as an export folds them); a real sherpa-onnx export names its
 * initializers by graph node, so binding one would need a name map this build does not have: create() rejects files without these names and says so.
 *
 * Deviations a deterministic checker needs (all in both backends): the source's additive noise comes from a counter-based hash (uniform, unit variance) instead of
 * torch.randn; dropout is inference-mode identity.  The harmonic phases follow the published SineGen._f02sine law evaluated in f64 (round 5; rounds 3-4 accumulated a per-sample
 * phase at piecewise-constant F0, which tests/kokoro_torch_ref.py — an independent restatement of the published modules — showed to be a different signal): per harmonic h and
 * F0 value m, r = frac(h f0_m / 24000); C_m = sum_{j <= m} r_j (mod 1); sample n sits at x = max(0, (n + 0.5) / 300 - 0.5) between F0 values m0 = floor(x) and
 * m1 = min(m0 + 1, M - 1): cycles = 300 (C_m0 + (x - m0) (m1 > m0 ? r_m1 : 0)) — torch's linear interpolation (align_corners = False) of the frame-rate phase, times the
 * up-sampling factor; sine = 0.1 sin(2 pi cycles).  (The published code adds torch.rand initial phases to the FIRST SAMPLE of the sample-rate phase increments, which its
 * own down-sampling to the frame rate never reads: they have no effect, and there are none here.  torch evaluates the law in f32, where a phase of 1e5 rad has an ulp of
 * 0.008 rad; f64 is the value the formula defines.)
 *
 * OPERATOR ARITHMETIC (the contract both backends implement; activations are row-major [time][channel] f32)
 *   contraction      every weight product is ONE f32 chain acc = fmaf(w, x, acc) from 0 in ascending k, bias added after; conv k = tap * Cin + ci
 *                    (GPU: v_mfma_f32_16x16x4_f32, which IS that chain bit for bit; CPU: fmaf) — conv / linear / LSTM / attention alike
 *   statistics       LayerNorm / AdaLN (over channels) and instance norm (over time): mean and centred variance in f64, rstd = (float)(1 / sqrt(var + (double)eps));
 *                    y = ((x - mean) * rstd) * g + b, AdaLN / AdaIN g = 1 + gamma(style), b = beta(style); softmax and duration sums in f64.  The ORDER of the f64 sums is part of
 *                    the contract, so that both backends round to the same f32: over channels and softmax columns "sum256" (256 lanes, lane l takes elements l, l + 256, ...
 *                    ascending; lanes combine by a butterfly xor 32 .. 1 within groups of 64, the four group sums add ascending — a 256-thread workgroup's reduction), over
 *                    time "chunk512" (runs of 512 steps sequentially, run sums ascending); the 50 duration bins sequentially
 *   transcendentals  exp through skw_expf (include/skw_math.h, bit-identical on both sides): sigmoid = 1 / (1 + exp(-x)), tanh = 1 - 2 / (exp(2x) + 1),
 *                    gelu_new = 0.5 x (1 + tanh(0.79788456 (x + 0.044715 x^3))); sin / cos / atan2 are the platform's (Snake, source, STFT phase,
 *                    inverse STFT; the DFT twiddles and the Hann window are table constants): a few ulps apart,
 *                    which is what the waveform tolerance in tests/test_gpu_kokoro.py covers
 */
#ifndef SKW_KOKORO_NET_H
#define SKW_KOKORO_NET_H
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace skw { namespace kokoro {

enum { STYLE_DIM = 128, N_FFT = 20, HOP = 5, N_BINS = 11, N_HARM = 9, SAMPLE_RATE = 24000, UPS0 = 10, UPS1 = 6, SRC_UP = UPS0 * UPS1 * HOP /* 300 samples per F0 value */, MAX_TOKENS = 510 };
enum Act { ACT_LEAKY02 = 0, ACT_LEAKY01 = 1, ACT_LEAKY001 = 2, ACT_GELU = 3, ACT_SNAKE = 4 };

/* a weight as the backends see it: host copy of the file's tensor + whatever the backend attached (device pointer, packed image) */
struct Tensor { std::vector<int64_t> dims; std::vector<float> host; void* dev = nullptr; void* packed = nullptr; void* packed2 = nullptr; long n() const { return (long)host.size(); } };
typedef std::map<std::string, Tensor> Weights;

/* geometry, derived from tensor shapes (check() fills it and names the first missing / mis-shaped tensor) */
struct Dims {
    int n_sym = 0, emb = 0, hid = 0, ffn = 0, heads = 0, n_layers = 12, max_pos = 0;      /* ALBERT */
    int d = 0, H = 0;                                                                     /* model width (512) and LSTM hidden per direction (256) */
    int max_dur = 0, te_depth = 0, te_k = 5;
    int dec_c = 0, asr_c = 0, gen_c0 = 0, gen_c1 = 0, gen_c2 = 0;                          /* decoder width (1024), asr_res (64), generator 512 -> 256 -> 128 */
    int n_decode = 0;
    int rk[3] = {3, 7, 11};
};

inline const Tensor* find(const Weights& w, const std::string& name) { auto it = w.find(name); return it == w.end() ? nullptr : &it->second; }

inline bool check(const Weights& w, Dims* g, std::string* err) {
    auto need = [&](const std::string& name, std::initializer_list<int64_t> shape) -> const Tensor* {
        const Tensor* t = find(w, name);
        if (!t) { *err = "model file: no tensor '" + name + "' (this build binds Kokoro's weights by PyTorch module name, include/skw_kokoro_net.h; a sherpa-onnx export names them by graph node)";
        return nullptr; }
        if (shape.size()) {
            bool ok = t->dims.size() == shape.size(); size_t i = 0;
            for (int64_t s : shape) { if (ok && s >= 0 && t->dims[i] != s) ok = false; ++i; }
            if (!ok) { *err = "model file: tensor '" + name + "' has an unexpected shape"; return nullptr; }
        }
        return t;
    };
    const Tensor* t;
    if (!(t = need("bert.embeddings.word_embeddings.weight", {-1, -1}))) return false;
    g->n_sym = (int)t->dims[0];
    g->emb = (int)t->dims[1];
    if (!(t = need("bert.embeddings.position_embeddings.weight", {-1, g->emb}))) return false;
    g->max_pos = (int)t->dims[0];
    if (!need("bert.embeddings.token_type_embeddings.weight", {-1, g->emb}) || !need("bert.embeddings.LayerNorm.weight", {g->emb}) || !need("bert.embeddings.LayerNorm.bias", {g->emb})) return false;
    if (!(t = need("bert.encoder.embedding_hidden_mapping_in.weight", {-1, g->emb}))) return false;
    g->hid = (int)t->dims[0];
    if (g->hid % 64) { *err = "model file: ALBERT hidden size must be a multiple of the 64-wide heads"; return false; }
    g->heads = g->hid / 64;
    const std::string L = "bert.encoder.albert_layer_groups.0.albert_layers.0.";
    for (const char* p : {"attention.query", "attention.key", "attention.value", "attention.dense"}) if (!need(L + p + ".weight", {g->hid, g->hid}) || !need(L + p + ".bias", {g->hid})) return false;
    if (!need(L + "attention.LayerNorm.weight", {g->hid}) || !need(L + "attention.LayerNorm.bias", {g->hid})) return false;
    if (!(t = need(L + "ffn.weight", {-1, g->hid}))) return false;
    g->ffn = (int)t->dims[0];
    if (!need(L + "ffn.bias", {g->ffn}) || !need(L + "ffn_output.weight", {g->hid, g->ffn}) || !need(L + "ffn_output.bias", {g->hid})) return false;
    if (!need(L + "full_layer_layer_norm.weight", {g->hid}) || !need(L + "full_layer_layer_norm.bias", {g->hid})) return false;
    if (const Tensor* nl = find(w, "bert.config.num_hidden_layers")) g->n_layers = (int)nl->host[0];      /* (a 1-element tensor the synthetic writer adds; 12 otherwise) */
    if (!(t = need("bert_encoder.weight", {-1, g->hid}))) return false;
    g->d = (int)t->dims[0];
    g->H = g->d / 2;
    if (!need("bert_encoder.bias", {g->d})) return false;
    if (g->d % 4 || g->H % 4) { *err = "model file: model width must be a multiple of 8"; return false; }
    auto lstm = [&](const std::string& p, int in) {
        for (const char* sfx : {"", "_reverse"})
            if (!need(p + "weight_ih_l0" + sfx, {4 * g->H, in}) || !need(p + "weight_hh_l0" + sfx, {4 * g->H, g->H}) || !need(p + "bias_ih_l0" + sfx, {4 * g->H})
                || !need(p + "bias_hh_l0" + sfx, {4 * g->H})) return false;
        return true;
    };
    for (int i = 0; i < 3; ++i) {
        if (!lstm("predictor.text_encoder.lstms." + std::to_string(2 * i) + ".", g->d + STYLE_DIM)) return false;
        const std::string a = "predictor.text_encoder.lstms." + std::to_string(2 * i + 1) + ".fc.";
        if (!need(a + "weight", {2 * g->d, STYLE_DIM}) || !need(a + "bias", {2 * g->d})) return false;
    }
    if (!lstm("predictor.lstm.", g->d + STYLE_DIM) || !lstm("predictor.shared.", g->d + STYLE_DIM)) return false;
    if (!(t = need("predictor.duration_proj.linear_layer.weight", {-1, g->d}))) return false;
    g->max_dur = (int)t->dims[0];
    if (!need("predictor.duration_proj.linear_layer.bias", {g->max_dur})) return false;
    auto resblk = [&](const std::string& p, int cin, int cout, bool up) {
        if (!need(p + "norm1.fc.weight", {2 * cin, STYLE_DIM}) || !need(p + "norm1.fc.bias", {2 * cin}) || !need(p + "norm2.fc.weight", {2 * cout, STYLE_DIM})
            || !need(p + "norm2.fc.bias", {2 * cout})) return false;
        if (!need(p + "conv1.weight", {cout, cin, 3}) || !need(p + "conv1.bias", {cout}) || !need(p + "conv2.weight", {cout, cout, 3}) || !need(p + "conv2.bias", {cout})) return false;
        if (cin != cout && !need(p + "conv1x1.weight", {cout, cin, 1})) return false;
        if (up && (!need(p + "pool.weight", {cin, 1, 3}) || !need(p + "pool.bias", {cin}))) return false;
        return true;
    };
    for (const char* br : {"predictor.F0.", "predictor.N."}) {
        const std::string b = br;
        if (!resblk(b + "0.", g->d, g->d, false) || !resblk(b + "1.", g->d, g->d / 2, true) || !resblk(b + "2.", g->d / 2, g->d / 2, false)) return false;
    }
    if (!need("predictor.F0_proj.weight", {1, g->d / 2, 1}) || !need("predictor.F0_proj.bias", {1}) || !need("predictor.N_proj.weight", {1, g->d / 2, 1})
        || !need("predictor.N_proj.bias", {1})) return false;
    if (!need("text_encoder.embedding.weight", {g->n_sym, g->d})) return false;
    g->te_depth = 0;
    while (find(w, "text_encoder.cnn." + std::to_string(g->te_depth) + ".0.weight")) ++g->te_depth;
    if (g->te_depth < 1) { *err = "model file: no tensor 'text_encoder.cnn.0.0.weight'"; return false; }
    for (int i = 0; i < g->te_depth; ++i) {
        const std::string p = "text_encoder.cnn." + std::to_string(i) + ".";
        if (!(t = need(p + "0.weight", {g->d, g->d, -1}))) return false;
        g->te_k = (int)t->dims[2];
        if (!need(p + "0.bias", {g->d}) || !need(p + "1.gamma", {g->d}) || !need(p + "1.beta", {g->d})) return false;
    }
    if (!lstm("text_encoder.lstm.", g->d)) return false;
    if (!(t = need("decoder.asr_res.0.weight", {-1, g->d, 1}))) return false;
    g->asr_c = (int)t->dims[0];
    if (!need("decoder.asr_res.0.bias", {g->asr_c}) || !need("decoder.F0_conv.weight", {1, 1, 3}) || !need("decoder.F0_conv.bias", {1}) || !need("decoder.N_conv.weight",
        {1, 1, 3}) || !need("decoder.N_conv.bias", {1})) return false;
    if (!(t = need("decoder.encode.conv1.weight", {-1, g->d + 2, 3}))) return false;
    g->dec_c = (int)t->dims[0];
    if (!resblk("decoder.encode.", g->d + 2, g->dec_c, false)) return false;
    g->n_decode = 0;
    while (find(w, "decoder.decode." + std::to_string(g->n_decode) + ".conv1.weight")) ++g->n_decode;
    if (g->n_decode < 1) { *err = "model file: no tensor 'decoder.decode.0.conv1.weight'"; return false; }
    if (!(t = need("decoder.generator.ups.0.weight", {-1, -1, 2 * UPS0}))) return false;
    g->gen_c0 = (int)t->dims[0];
    g->gen_c1 = (int)t->dims[1];
    const int cat = g->dec_c + 2 + g->asr_c;
    for (int i = 0; i < g->n_decode; ++i) {
        const bool last = i + 1 == g->n_decode;
        if (!resblk("decoder.decode." + std::to_string(i) + ".", cat, last ? g->gen_c0 : g->dec_c, last)) return false;
    }
    if (!need("decoder.generator.ups.0.bias", {g->gen_c1})) return false;
    if (!(t = need("decoder.generator.ups.1.weight", {g->gen_c1, -1, 2 * UPS1}))) return false;
    g->gen_c2 = (int)t->dims[1];
    if (!need("decoder.generator.ups.1.bias", {g->gen_c2})) return false;
    if (!need("decoder.generator.m_source.l_linear.weight", {1, N_HARM}) || !need("decoder.generator.m_source.l_linear.bias", {1})) return false;
    if (!need("decoder.generator.noise_convs.0.weight", {g->gen_c1, 2 * N_BINS, 2 * UPS1}) || !need("decoder.generator.noise_convs.0.bias", {g->gen_c1})) return false;
    if (!need("decoder.generator.noise_convs.1.weight", {g->gen_c2, 2 * N_BINS, 1}) || !need("decoder.generator.noise_convs.1.bias", {g->gen_c2})) return false;
    auto adain1 = [&](const std::string& p, int c, int k) {
        for (int j = 0; j < 3; ++j) {
            const std::string J = std::to_string(j);
            if (!need(p + "convs1." + J + ".weight", {c, c, k}) || !need(p + "convs1." + J + ".bias", {c}) || !need(p + "convs2." + J + ".weight", {c, c, k})
                || !need(p + "convs2." + J + ".bias", {c})) return false;
            if (!need(p + "adain1." + J + ".fc.weight", {2 * c, STYLE_DIM}) || !need(p + "adain1." + J + ".fc.bias", {2 * c}) || !need(p + "adain2." + J + ".fc.weight",
                {2 * c, STYLE_DIM}) || !need(p + "adain2." + J + ".fc.bias", {2 * c})) return false;
            if (!need(p + "alpha1." + J, {c}) || !need(p + "alpha2." + J, {c})) return false;
        }
        return true;
    };
    if (!adain1("decoder.generator.noise_res.0.", g->gen_c1, 7) || !adain1("decoder.generator.noise_res.1.", g->gen_c2, 11)) return false;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) if (!adain1("decoder.generator.resblocks." + std::to_string(3 * i + j) + ".", i ? g->gen_c2 : g->gen_c1, g->rk[j])) return false;
    if (!need("decoder.generator.conv_post.weight", {2 * N_BINS, g->gen_c2, 7}) || !need("decoder.generator.conv_post.bias", {2 * N_BINS})) return false;
    return true;
}

/* what a forward pass leaves for the caller and the parity taps */
template <class Buf> struct Outputs {
    std::vector<int> dur;      /* per token */
    int F = 0;                 /* frames = sum of durations; F0 / N curves have 2 F values, the waveform 600 F samples */
    Buf bert, d_en, t_en, f0, n, dec, har, post, audio;
};

/*
 * The backend B provides (W = const Tensor&, W* = const Tensor*):
 *   typedef Buf (rows T, channels C, data);  Buf copy(Buf);  Buf embed(W table, ids, T);  Buf add_pos_type(Buf x, W pos, W type);
 *   Buf conv(Buf x, W w, W* bias, int K, int stride, int dil, int pad)      [out rows = (T + 2 pad - dil (K - 1) - 1) / stride + 1];
 *   Buf convtr(Buf x, W w, W* bias, int K, int stride, int pad, int out_pad, bool depthwise);
 *   void layernorm(Buf&, W gamma, W beta, float eps);  Buf style_fc(W w, W b, const float* style)      [a 1-row Buf];
 *   void ada_ln(Buf& x, Buf gb)      (LayerNorm over channels, y = n (1 + gb[c]) + gb[C + c]);  void ada_in_act(Buf& x, Buf gb, Act, W* alpha)      (instance norm over time, same affine, then
 *   the activation: the normalised value is rounded to f32 before the activation takes it, exactly as two passes would; a backend may make it one);
 *   void add_scale(Buf& a, Buf b, float f)      ((a + b) * f);
 *   void act(Buf&, Act, W* alpha);  Buf concat(std::vector<Buf>)      (channels);  Buf concat_style(Buf x, const float* style);  void add(Buf& a, Buf b);  void scale(Buf&, float);
 *   Buf upsample2(Buf);  Buf reflect_pad_left(Buf);  Buf attention(Buf q, Buf k, Buf v, int heads);  Buf lstm_bi(Buf x, W* [8] weights fwd / rev);
 *   Buf gather_rows(Buf x, const std::vector<int>& rows);  std::vector<int> durations(Buf logits, float scale);
 *   Buf source_stft(Buf f0curve, W lin_w, W lin_b);  Buf istft(Buf post)
 * `style` is whatever the backend's operators read: a host pointer for the CPU backend, a device pointer for the GPU's.
 */
template <class B> struct Net {
    typedef typename B::Buf Buf;
    B& be; const Weights& w; Dims g;
    Net(B& be_, const Weights& w_, const Dims& g_) : be(be_), w(w_), g(g_) {}
    const Tensor& W(const std::string& n) const { return w.at(n); }
    const Tensor* Wopt(const std::string& n) const { return find(w, n); }

    Buf linear(const Buf& x, const std::string& p) { return be.conv(x, W(p + ".weight"), Wopt(p + ".bias"), 1, 1, 1, 0); }
    Buf lstm(const Buf& x, const std::string& p) {
        const Tensor* ws[8] = {&W(p + "weight_ih_l0"), &W(p + "weight_hh_l0"), &W(p + "bias_ih_l0"), &W(p + "bias_hh_l0"),
                               &W(p + "weight_ih_l0_reverse"), &W(p + "weight_hh_l0_reverse"), &W(p + "bias_ih_l0_reverse"), &W(p + "bias_hh_l0_reverse")};
        return be.lstm_bi(x, ws);
    }
    /* AdainResBlk1d: out = (residual(x) + shortcut(x)) / sqrt 2 */
    Buf adain_resblk(const Buf& x, const std::string& p, const float* style, bool up) {
        const Tensor* c1x1 = Wopt(p + "conv1x1.weight");
        Buf sc = up ? be.upsample2(x) : x;
        if (c1x1) sc = be.conv(sc, *c1x1, nullptr, 1, 1, 1, 0);
        Buf r = be.copy(x);
        be.ada_in_act(r, be.style_fc(W(p + "norm1.fc.weight"), W(p + "norm1.fc.bias"), style), ACT_LEAKY02, nullptr);
        if (up) r = be.convtr(r, W(p + "pool.weight"), Wopt(p + "pool.bias"), 3, 2, 1, 1, true);
        r = be.conv(r, W(p + "conv1.weight"), Wopt(p + "conv1.bias"), 3, 1, 1, 1);
        be.ada_in_act(r, be.style_fc(W(p + "norm2.fc.weight"), W(p + "norm2.fc.bias"), style), ACT_LEAKY02, nullptr);
        r = be.conv(r, W(p + "conv2.weight"), Wopt(p + "conv2.bias"), 3, 1, 1, 1);
        be.add_scale(r, sc, 0.70710678118654752440f);
        return r;
    }
    /* AdaINResBlock1 (generator): three [AdaIN, Snake, dilated conv, AdaIN, Snake, conv] residual steps */
    Buf adain_resblock1(const Buf& x_in, const std::string& p, const float* style, int k) {
        static const int dil[3] = {1, 3, 5};
        Buf x = be.copy(x_in);
        for (int j = 0; j < 3; ++j) {
            const std::string J = std::to_string(j);
            Buf t = be.copy(x);
            be.ada_in_act(t, be.style_fc(W(p + "adain1." + J + ".fc.weight"), W(p + "adain1." + J + ".fc.bias"), style), ACT_SNAKE, &W(p + "alpha1." + J));
            t = be.conv(t, W(p + "convs1." + J + ".weight"), Wopt(p + "convs1." + J + ".bias"), k, 1, dil[j], dil[j] * (k - 1) / 2);
            be.ada_in_act(t, be.style_fc(W(p + "adain2." + J + ".fc.weight"), W(p + "adain2." + J + ".fc.bias"), style), ACT_SNAKE, &W(p + "alpha2." + J));
            t = be.conv(t, W(p + "convs2." + J + ".weight"), Wopt(p + "convs2." + J + ".bias"), k, 1, 1, (k - 1) / 2);
            be.add(t, x);
            x = t;
        }
        return x;
    }

    /* ids: token ids incl. the pad id 0 at both ends; style: the voice's 256-float row (acoustic half first); scale = length_scale / speed */
    bool forward(const std::vector<int>& ids, const float* style, float scale, int max_frames, Outputs<Buf>* out, std::string* err) {
        const int T = (int)ids.size();
        if (T < 1 || T > g.max_pos || T > MAX_TOKENS) { *err = "token count outside the model's position table"; return false; }
        for (int id : ids) if (id < 0 || id >= g.n_sym) { *err = "token id outside the embedding table"; return false; }
        const float* s_ac = style; const float* s_pr = style + STYLE_DIM;
        /* ---- bert (ALBERT, one shared layer applied n_layers times) ---- */
        Buf x = be.embed(W("bert.embeddings.word_embeddings.weight"), ids.data(), T);
        x = be.add_pos_type(x, W("bert.embeddings.position_embeddings.weight"), W("bert.embeddings.token_type_embeddings.weight"));
        be.layernorm(x, W("bert.embeddings.LayerNorm.weight"), W("bert.embeddings.LayerNorm.bias"), 1e-12f);
        x = linear(x, "bert.encoder.embedding_hidden_mapping_in");
        const std::string L = "bert.encoder.albert_layer_groups.0.albert_layers.0.";
        for (int l = 0; l < g.n_layers; ++l) {
            Buf a = be.attention(linear(x, L + "attention.query"), linear(x, L + "attention.key"), linear(x, L + "attention.value"), g.heads);
            a = linear(a, L + "attention.dense");
            be.add(a, x);
            be.layernorm(a, W(L + "attention.LayerNorm.weight"), W(L + "attention.LayerNorm.bias"), 1e-12f);
            Buf f = linear(a, L + "ffn");
            be.act(f, ACT_GELU, nullptr);
            f = linear(f, L + "ffn_output");
            be.add(f, a);
            be.layernorm(f, W(L + "full_layer_layer_norm.weight"), W(L + "full_layer_layer_norm.bias"), 1e-12f);
            x = f;
        }
        out->bert = x;
        Buf d_en = linear(x, "bert_encoder");
        out->d_en = d_en;
        /* ---- predictor: DurationEncoder, durations ---- */
        Buf d = be.concat_style(d_en, s_pr);
        for (int i = 0; i < 3; ++i) {
            Buf h = lstm(d, "predictor.text_encoder.lstms." + std::to_string(2 * i) + ".");
            const std::string a = "predictor.text_encoder.lstms." + std::to_string(2 * i + 1) + ".fc.";
            be.ada_ln(h, be.style_fc(W(a + "weight"), W(a + "bias"), s_pr));
            d = be.concat_style(h, s_pr);
        }
        Buf xl = lstm(d, "predictor.lstm.");
        Buf logits = linear(xl, "predictor.duration_proj.linear_layer");
        out->dur = be.durations(logits, scale);
        std::vector<int> rows; rows.reserve(1024);
        for (int t = 0; t < T; ++t) for (int k = 0; k < out->dur[t]; ++k) rows.push_back(t);
        const int F = (int)rows.size();
        out->F = F;
        if (F < 1 || F > max_frames) { *err = "Generated audio too long (" + std::to_string(F) + " frames; at most " + std::to_string(max_frames) + ")"; return false; }
        /* ---- predictor: F0 / N curves over frames ---- */
        Buf en = be.gather_rows(d, rows);
        Buf sh = lstm(en, "predictor.shared.");
        Buf curves[2];
        for (int b = 0; b < 2; ++b) {
            const std::string p = b ? "predictor.N." : "predictor.F0.";
            Buf c = adain_resblk(sh, p + "0.", s_pr, false);
            c = adain_resblk(c, p + "1.", s_pr, true);
            c = adain_resblk(c, p + "2.", s_pr, false);
            curves[b] = be.conv(c, W(b ? "predictor.N_proj.weight" : "predictor.F0_proj.weight"), Wopt(b ? "predictor.N_proj.bias" : "predictor.F0_proj.bias"), 1, 1, 1, 0);      /* [2 F][1] */
        }
        out->f0 = curves[0]; out->n = curves[1];
        /* ---- text encoder (acoustic) ---- */
        Buf te = be.embed(W("text_encoder.embedding.weight"), ids.data(), T);
        for (int i = 0; i < g.te_depth; ++i) {
            const std::string p = "text_encoder.cnn." + std::to_string(i) + ".";
            te = be.conv(te, W(p + "0.weight"), Wopt(p + "0.bias"), g.te_k, 1, 1, (g.te_k - 1) / 2);
            be.layernorm(te, W(p + "1.gamma"), W(p + "1.beta"), 1e-5f);
            be.act(te, ACT_LEAKY02, nullptr);
        }
        te = lstm(te, "text_encoder.lstm.");
        out->t_en = te;
        Buf asr = be.gather_rows(te, rows);
        /* ---- decoder ---- */
        Buf f0d = be.conv(curves[0], W("decoder.F0_conv.weight"), Wopt("decoder.F0_conv.bias"), 3, 2, 1, 1);      /* [F][1] */
        Buf nd = be.conv(curves[1], W("decoder.N_conv.weight"), Wopt("decoder.N_conv.bias"), 3, 2, 1, 1);
        Buf xd = adain_resblk(be.concat({asr, f0d, nd}), "decoder.encode.", s_ac, false);
        Buf asr_res = be.conv(asr, W("decoder.asr_res.0.weight"), Wopt("decoder.asr_res.0.bias"), 1, 1, 1, 0);
        for (int i = 0; i < g.n_decode; ++i)
            xd = adain_resblk(be.concat({xd, asr_res, f0d, nd}), "decoder.decode." + std::to_string(i) + ".", s_ac, i + 1 == g.n_decode);      /* the last block doubles the frame rate */
        out->dec = xd;                                                                                                                       /* [2 F][gen_c0] */
        /* ---- generator (ISTFTNet) ---- */
        Buf har = be.source_stft(curves[0], W("decoder.generator.m_source.l_linear.weight"), W("decoder.generator.m_source.l_linear.bias"));   /* [2 F * 60 + 1][22] */
        out->har = har;
        Buf gx = be.copy(xd);      /* (the activations below work in place: the decoder-output tap must not see them — round 5: the independent torch checker found the tap post-LeakyReLU) */
        for (int i = 0; i < 2; ++i) {
            const std::string I = std::to_string(i);
            be.act(gx, ACT_LEAKY01, nullptr);
            Buf xs = i == 0 ? be.conv(har, W("decoder.generator.noise_convs.0.weight"), Wopt("decoder.generator.noise_convs.0.bias"), 2 * UPS1, UPS1, 1, UPS1 / 2)
                            : be.conv(har, W("decoder.generator.noise_convs.1.weight"), Wopt("decoder.generator.noise_convs.1.bias"), 1, 1, 1, 0);
            xs = adain_resblock1(xs, "decoder.generator.noise_res." + I + ".", s_ac, i ? 11 : 7);
            const int up = i ? UPS1 : UPS0;
            gx = be.convtr(gx, W("decoder.generator.ups." + I + ".weight"), Wopt("decoder.generator.ups." + I + ".bias"), 2 * up, up, up / 2, 0, false);
            if (i == 1) gx = be.reflect_pad_left(gx);
            be.add(gx, xs);
            Buf sum = adain_resblock1(gx, "decoder.generator.resblocks." + std::to_string(3 * i) + ".", s_ac, g.rk[0]);
            be.add(sum, adain_resblock1(gx, "decoder.generator.resblocks." + std::to_string(3 * i + 1) + ".", s_ac, g.rk[1]));
            be.add_scale(sum, adain_resblock1(gx, "decoder.generator.resblocks." + std::to_string(3 * i + 2) + ".", s_ac, g.rk[2]), 1.0f / 3.0f);
            gx = sum;
        }
        be.act(gx, ACT_LEAKY001, nullptr);
        Buf post = be.conv(gx, W("decoder.generator.conv_post.weight"), Wopt("decoder.generator.conv_post.bias"), 7, 1, 1, 3);      /* [2 F * 60 + 1][22] */
        out->post = post;
        out->audio = be.istft(post);                                                                                                /* [600 F][1] */
        return true;
    }
};

/* ---- the 20-point DFT's twiddles cos / sin(2 pi j / 20), j = 0 .. 19, as exact constants with the circle's symmetries (so that sin(pi m) IS 0: with the platform's sin the Nyquist bin's
 * imaginary part is rounding noise whose sign, hence a +-pi flip of that bin's phase, differs between libm and the device library); each backend declares its own array from these ---- */
#define SKW_KOKORO_TW_A 0.9510565162951535
#define SKW_KOKORO_TW_B 0.8090169943749475
#define SKW_KOKORO_TW_C 0.5877852522924731
#define SKW_KOKORO_TW_D 0.30901699437494745
#define SKW_KOKORO_TW_COS {1.0, SKW_KOKORO_TW_A, SKW_KOKORO_TW_B, SKW_KOKORO_TW_C, SKW_KOKORO_TW_D, 0.0, -SKW_KOKORO_TW_D, -SKW_KOKORO_TW_C, -SKW_KOKORO_TW_B, -SKW_KOKORO_TW_A, \
                           -1.0, -SKW_KOKORO_TW_A, -SKW_KOKORO_TW_B, -SKW_KOKORO_TW_C, -SKW_KOKORO_TW_D, 0.0, SKW_KOKORO_TW_D, SKW_KOKORO_TW_C, SKW_KOKORO_TW_B, SKW_KOKORO_TW_A}
#define SKW_KOKORO_TW_SIN {0.0, SKW_KOKORO_TW_D, SKW_KOKORO_TW_C, SKW_KOKORO_TW_B, SKW_KOKORO_TW_A, 1.0, SKW_KOKORO_TW_A, SKW_KOKORO_TW_B, SKW_KOKORO_TW_C, SKW_KOKORO_TW_D, \
                           0.0, -SKW_KOKORO_TW_D, -SKW_KOKORO_TW_C, -SKW_KOKORO_TW_B, -SKW_KOKORO_TW_A, -1.0, -SKW_KOKORO_TW_A, -SKW_KOKORO_TW_B, -SKW_KOKORO_TW_C, -SKW_KOKORO_TW_D}

/* ---- pieces of the source that both backends compute with the same integer arithmetic ---- */
/* uniform in [-sqrt 3, sqrt 3): zero mean, unit variance (stands in for torch.randn: the reference's noise is not reproducible either) */
#if defined(__HIPCC__)
__host__ __device__
#endif
inline float unit_noise(uint64_t counter) {
    uint64_t x = counter + 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
    const uint32_t u = (uint32_t)(x >> 40);                                   /* 24 bits */
    return ((float)u * (1.0f / 8388608.0f) - 1.0f) * 1.7320508075688772f;
}

}}  // namespace skw::kokoro
#endif
