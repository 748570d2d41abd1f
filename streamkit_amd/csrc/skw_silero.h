// skw_silero.h — Silero VAD (v5/v6 graph, 16 kHz branch) evaluated on the CPU inside the plugin: SURVEY.md §8a row W2 / §8f-2.
//
// Replaces what the reference reaches through `ort` (onnxruntime): /root/reference/plugins/native/whisper/src/vad.rs
//   :34-55  SileroVAD::new       -> SileroVad::load      (reads the .onnx file named by `vad_model_path`)
//   :67-120 process_chunk        -> SileroVad::process_chunk: input [1, 576] = 64 context samples + the 512-sample frame,
//                                   state [2, 1, 128] (h, c) carried between calls, sr = 16000; returns the speech probability;
//                                   context <- last 64 samples of the frame
//   :139-142 reset               -> SileroVad::reset
// It is sequential per stream (an LSTM, ~31 calls per audio-second, ~0.4 MFLOP each) and is therefore CPU code by design.
//
// The ONNX runtime and the model file are third-party and absent offline (parity unpinned).  What is restated here
// is the published Silero v5 graph as recalled: reflect-pad 64 on the right -> STFT as a strided Conv1d with the stored basis
// (n_fft 256, hop 128, 129 bins) -> magnitude -> four Conv1d+ReLU blocks (129->128 s1, 128->64 s2, 64->64 s2, 64->128 s1,
// kernel 3, padding 1) -> LSTMCell(128, 128) -> ReLU -> Conv1d(128 -> 1, k = 1) -> sigmoid.  Weights are located in the file by
// SHAPE inside the (sub)graph that holds the 16 kHz STFT basis [258, 1, 256], not by node names, because the names of the
// exported graph could not be inspected here; tools/make_synth_silero.py writes files of the same structure (an `If` node on the
// sample rate with one sub-graph per rate) with seeded weights, and oracle/skw_silero_oracle.c restates the arithmetic
// independently (tests/test_cpu_silero.py also checks both against a torch.nn.functional restatement).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <string>
#include <vector>

namespace skw {
namespace onnx {

// ---- protobuf wire format, the subset ONNX files use (varint, 64-bit, length-delimited, 32-bit)
struct Buf { const uint8_t* p; const uint8_t* e; };
inline bool varint(Buf& b, uint64_t* v) {
    uint64_t r = 0; int sh = 0;
    while (b.p < b.e && sh < 64) { const uint8_t c = *b.p++; r |= (uint64_t)(c & 0x7f) << sh; if (!(c & 0x80)) { *v = r; return true; } sh += 7; }
    return false;
}
struct Field { uint32_t num = 0, wt = 0; uint64_t val = 0; Buf sub{nullptr, nullptr}; };
inline bool next_field(Buf& b, Field* f) {
    uint64_t key; if (!varint(b, &key)) return false;
    f->num = (uint32_t)(key >> 3); f->wt = (uint32_t)(key & 7);
    switch (f->wt) {
        case 0: return varint(b, &f->val);
        case 1: if (b.e - b.p < 8) return false; memcpy(&f->val, b.p, 8); b.p += 8; return true;
        case 2: { uint64_t n; if (!varint(b, &n) || (uint64_t)(b.e - b.p) < n) return false; f->sub.p = b.p; f->sub.e = b.p + n; b.p += n; return true; }
        case 5: { if (b.e - b.p < 4) return false; uint32_t v; memcpy(&v, b.p, 4); f->val = v; b.p += 4; return true; }
        default: return false;
    }
}
inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t s = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1f; uint32_t m = h & 0x3ff, o;
    if (e == 0) { if (!m) o = s; else { uint32_t sh = 0; while (!(m & 0x400)) { m <<= 1; sh++; } o = s | ((113 - sh) << 23) | ((m & 0x3ff) << 13); } }
    else if (e == 31) o = s | 0x7f800000u | (m << 13); else o = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &o, 4); return f;
}

struct Tensor { std::string name; std::vector<int64_t> dims; std::vector<float> data; int graph = 0; int order = 0; int data_type = 0;
                size_t numel() const { size_t n = 1; for (int64_t d : dims) n *= (size_t)d; return n; }
                bool is(std::initializer_list<int64_t> s) const { return dims.size() == s.size() && std::equal(dims.begin(), dims.end(), s.begin()); } };

// TensorProto: 1 dims, 2 data_type, 4 float_data, 5 int32_data (f16 bits), 8 name, 9 raw_data, 10 double_data, 14 data_location
inline bool parse_tensor(Buf b, Tensor* t, std::string* err) {
    std::vector<float> fdata; std::vector<uint16_t> hdata; std::vector<double> ddata; Buf raw{nullptr, nullptr}; Field f;
    while (b.p < b.e) {
        if (!next_field(b, &f)) { *err = "corrupt TensorProto"; return false; }
        if (f.num == 1) { if (f.wt == 2) { Buf s = f.sub; uint64_t v; while (s.p < s.e) { if (!varint(s, &v)) { *err = "corrupt dims";
        return false; } t->dims.push_back((int64_t)v); } } else t->dims.push_back((int64_t)f.val); }
        else if (f.num == 2) t->data_type = (int)f.val;
        else if (f.num == 4) { if (f.wt == 2) { const size_t n = (f.sub.e - f.sub.p) / 4;
        const size_t o = fdata.size(); fdata.resize(o + n); memcpy(fdata.data() + o, f.sub.p, n * 4);
        } else { uint32_t v = (uint32_t)f.val; float x; memcpy(&x, &v, 4); fdata.push_back(x); } }
        else if (f.num == 5) { if (f.wt == 2) { Buf s = f.sub; uint64_t v; while (s.p < s.e) { if (!varint(s, &v)) { *err = "corrupt int32_data";
        return false; } hdata.push_back((uint16_t)v); } } else hdata.push_back((uint16_t)f.val); }
        else if (f.num == 8 && f.wt == 2) t->name.assign((const char*)f.sub.p, f.sub.e - f.sub.p);
        else if (f.num == 9 && f.wt == 2) raw = f.sub;
        else if (f.num == 10 && f.wt == 2) { const size_t n = (f.sub.e - f.sub.p) / 8; ddata.resize(n); memcpy(ddata.data(), f.sub.p, n * 8); }
        else if (f.num == 14 && f.val == 1) { *err = "tensor '" + t->name + "' uses external data"; return false; }
    }
    // the dims come from the file: nothing is sized by them before they are known to be sane (a malformed or merely unusual .onnx must fail here
    // with a message, not as std::length_error / bad_alloc somewhere below)
    size_t n = 1;
    for (int64_t d : t->dims) {
        if (d < 0 || d > (int64_t)(64u << 20)) { *err = "tensor '" + t->name + "': implausible dimension"; return false; }
        n *= (size_t)d;
        if (n > (size_t)(64u << 20)) { *err = "tensor '" + t->name + "': more than 64M elements"; return false; }
    }
    if (t->data_type == 1) { if (raw.p) { if ((size_t)(raw.e - raw.p) != n * 4) { *err = "tensor '" + t->name + "': raw_data size";
    return false; } t->data.resize(n); memcpy(t->data.data(), raw.p, n * 4); } else t->data = fdata; }
    else if (t->data_type == 10) { t->data.resize(raw.p ? n : hdata.size());
    if (raw.p) { if ((size_t)(raw.e - raw.p) != n * 2) { *err = "tensor '" + t->name + "': raw_data size";
    return false; } for (size_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, raw.p + 2 * i, 2);
    t->data[i] = f16_bits_to_f32(h); } } else for (size_t i = 0; i < hdata.size(); ++i) t->data[i] = f16_bits_to_f32(hdata[i]); }
    else if (t->data_type == 11) { if (raw.p) { ddata.resize(n); if ((size_t)(raw.e - raw.p) != n * 8) { *err = "tensor '" + t->name + "': raw_data size";
    return false; } memcpy(ddata.data(), raw.p, n * 8); } t->data.assign(ddata.begin(), ddata.end()); }
    // other element types (int64 shapes, bools) carry no weights: kept with empty data
    if (!t->data.empty() && t->data.size() != n) { *err = "tensor '" + t->name + "': element count does not match dims"; return false; }
    return true;
}

struct Model { std::vector<Tensor> tensors; int n_graphs = 0; };
inline bool parse_graph(Buf b, int gid, Model* m, std::string* err, int depth);
// NodeProto: 2 output, 4 op_type, 5 attribute { 1 name, 5 t, 6 g, 10 tensors, 11 graphs }
inline bool parse_node(Buf b, int gid, Model* m, std::string* err, int depth) {
    std::string out0; std::vector<Buf> attrs; Field f; bool have_out = false;
    while (b.p < b.e) {
        if (!next_field(b, &f)) { *err = "corrupt NodeProto"; return false; }
        if (f.num == 2 && f.wt == 2 && !have_out) { out0.assign((const char*)f.sub.p, f.sub.e - f.sub.p); have_out = true; }
        else if (f.num == 5 && f.wt == 2) attrs.push_back(f.sub);
    }
    for (Buf a : attrs) {
        while (a.p < a.e) {
            if (!next_field(a, &f)) { *err = "corrupt AttributeProto"; return false; }
            if ((f.num == 5 || f.num == 10) && f.wt == 2) { Tensor t; t.graph = gid;
            t.order = (int)m->tensors.size(); if (!parse_tensor(f.sub, &t, err)) return false;
            if (t.name.empty()) t.name = out0; m->tensors.push_back(std::move(t)); }
            else if ((f.num == 6 || f.num == 11) && f.wt == 2) { if (depth > 8) { *err = "sub-graphs nested too deeply";
            return false; } const int sub = m->n_graphs++; if (!parse_graph(f.sub, sub, m, err, depth + 1)) return false; }
        }
    }
    return true;
}
// GraphProto: 1 node, 5 initializer
inline bool parse_graph(Buf b, int gid, Model* m, std::string* err, int depth) {
    Field f;
    while (b.p < b.e) {
        if (!next_field(b, &f)) { *err = "corrupt GraphProto"; return false; }
        if (f.num == 1 && f.wt == 2) { if (!parse_node(f.sub, gid, m, err, depth)) return false; }
        else if (f.num == 5 && f.wt == 2) { Tensor t; t.graph = gid; t.order = (int)m->tensors.size(); if (!parse_tensor(f.sub, &t, err)) return false; m->tensors.push_back(std::move(t)); }
    }
    return true;
}
// ModelProto: 7 graph
inline bool parse_model(const std::vector<uint8_t>& bytes, Model* m, std::string* err) {
    Buf b{bytes.data(), bytes.data() + bytes.size()}; Field f; bool seen = false;
    while (b.p < b.e) {
        if (!next_field(b, &f)) { *err = "not an ONNX ModelProto"; return false; }
        if (f.num == 7 && f.wt == 2) { seen = true; const int g = m->n_graphs++; if (!parse_graph(f.sub, g, m, err, 0)) return false; }
    }
    if (!seen) { *err = "no graph in the ONNX file"; return false; }
    return true;
}
}  // namespace onnx

// ------------------------------------------------------------------ the network
struct SileroWeights {
    std::vector<float> basis;                       // [258][256]
    std::vector<float> cw[4], cb[4];                // conv weights [co][ci][3], biases [co]
    std::vector<float> w_ih, w_hh, b_ih, b_hh;      // LSTMCell, PyTorch gate order i, f, g, o: [512][128], [512]
    std::vector<float> ow; float ob = 0.0f;         // final Conv1d(128 -> 1, k = 1)
    std::string bound;                              // which tensors of the file were bound to which role (logged by the plugin at INFO)
};
static const int SILERO_CI[4] = {129, 128, 64, 64}, SILERO_CO[4] = {128, 64, 64, 128}, SILERO_STRIDE[4] = {1, 2, 2, 1};

inline bool silero_bind(const onnx::Model& m, SileroWeights* w, std::string* err) {
    int g16 = -1;
    for (const auto& t : m.tensors) if (t.is({258, 1, 256}) && !t.data.empty()) { g16 = t.graph; break; }
    if (g16 < 0) { *err = "no 16 kHz STFT basis [258, 1, 256] in the model (is this a Silero VAD v5/v6 file?)"; return false; }
    std::vector<const onnx::Tensor*> ts; for (const auto& t : m.tensors) if (t.graph == g16 && !t.data.empty()) ts.push_back(&t);
    std::vector<char> used(ts.size(), 0);
    auto find = [&](std::initializer_list<int64_t> dims, const char* hint, size_t after) -> int {
        int first = -1;
        for (size_t i = 0; i < ts.size(); ++i) if (!used[i] && ts[i]->is(dims)) { if (hint && ts[i]->name.find(hint) != std::string::npos) return (int)i; if (first < 0 && i >= after) first = (int)i; }
        if (first < 0) for (size_t i = 0; i < ts.size(); ++i) if (!used[i] && ts[i]->is(dims)) return (int)i;
        return first;
    };
    auto take = [&](std::initializer_list<int64_t> dims, const char* hint, size_t after, std::vector<float>* dst, const char* what, size_t* pos) -> bool {
        const int i = find(dims, hint, after); if (i < 0) { *err = std::string("Silero VAD model: missing tensor ") + what; return false; }
        used[i] = 1; *dst = ts[i]->data; if (pos) *pos = (size_t)i;
        w->bound += (w->bound.empty() ? "" : ", ") + std::string(what) + " <- '" + ts[i]->name + "'"; return true;
    };
    if (!take({258, 1, 256}, nullptr, 0, &w->basis, "STFT basis [258,1,256]", nullptr)) return false;
    for (int l = 0; l < 4; ++l) {
        size_t pos = 0; char what[64]; snprintf(what, sizeof what, "encoder conv %d weight [%d,%d,3]", l, SILERO_CO[l], SILERO_CI[l]);
        if (!take({SILERO_CO[l], SILERO_CI[l], 3}, nullptr, 0, &w->cw[l], what, &pos)) return false;
        // the bias: same name stem when there is one, else the next unused vector of that length after the weight
        std::string stem = ts[pos]->name; const size_t k = stem.rfind("weight"); std::string bias_name = k != std::string::npos ? stem.substr(0, k) + "bias" : std::string();
        snprintf(what, sizeof what, "encoder conv %d bias [%d]", l, SILERO_CO[l]);
        if (!take({SILERO_CO[l]}, bias_name.empty() ? nullptr : bias_name.c_str(), pos, &w->cb[l], what, nullptr)) return false;
    }
    if (find({512, 128}, nullptr, 0) >= 0) {         // LSTMCell as exported by tracing: weight_ih, weight_hh [4H, H], bias_ih, bias_hh [4H]
        size_t p0 = 0, p1 = 0;
        // name hints first (PyTorch's weight_ih / weight_hh survive tracing), file order second
        if (!take({512, 128}, "ih", 0, &w->w_ih, "LSTM weight_ih [512,128]", &p0) || !take({512, 128}, "hh", p0, &w->w_hh, "LSTM weight_hh [512,128]", &p1)) return false;
        if (!take({512}, "ih", p0, &w->b_ih, "LSTM bias_ih [512]", nullptr) || !take({512}, "hh", p1, &w->b_hh, "LSTM bias_hh [512]", nullptr)) return false;
    } else {                                          // ONNX LSTM operator: W, R [1, 4H, H] and B [1, 8H], gate order i, o, f, c
        std::vector<float> W, R, Bv; size_t p0 = 0;
        if (!take({1, 512, 128}, "W", 0, &W, "LSTM W [1,512,128]", &p0) || !take({1, 512, 128}, "R", p0, &R, "LSTM R [1,512,128]", nullptr) || !take({1, 1024}, nullptr,
            0, &Bv, "LSTM B [1,1024]", nullptr)) return false;
        static const int from_iofc[4] = {0, 2, 3, 1};                       // PyTorch block (i, f, g, o) <- ONNX block (i, o, f, c)
        w->w_ih.resize(512 * 128); w->w_hh.resize(512 * 128); w->b_ih.resize(512); w->b_hh.resize(512);
        for (int gt = 0; gt < 4; ++gt) {
            const int src = from_iofc[gt];
            memcpy(&w->w_ih[(size_t)gt * 128 * 128], &W[(size_t)src * 128 * 128], sizeof(float) * 128 * 128);
            memcpy(&w->w_hh[(size_t)gt * 128 * 128], &R[(size_t)src * 128 * 128], sizeof(float) * 128 * 128);
            memcpy(&w->b_ih[gt * 128], &Bv[src * 128], sizeof(float) * 128); memcpy(&w->b_hh[gt * 128], &Bv[512 + src * 128], sizeof(float) * 128);
        }
    }
    std::vector<float> ob; size_t pos = 0;
    if (!take({1, 128, 1}, nullptr, 0, &w->ow, "output conv weight [1,128,1]", &pos) || !take({1}, nullptr, pos, &ob, "output conv bias [1]", nullptr)) return false;
    w->ob = ob[0];
    return true;
}

// One stream's Silero state.  process_chunk follows vad.rs:67-120 call for call.
class SileroVad {
public:
    // SileroVAD::new (vad.rs:34-55): only 16 kHz is used by the plugin (lib.rs:382); errors carry the "Failed to load VAD model from '<path>': ..." text
    static bool load_weights(const std::string& path, SileroWeights* w, std::string* err) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) { *err = "Failed to load VAD model from '" + path + "': cannot open file"; return false; }
        std::vector<uint8_t> bytes; uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) { bytes.insert(bytes.end(), buf, buf + n); if (bytes.size() > (64u << 20)) break; }
        fclose(f);
        onnx::Model m; std::string e;
        if (bytes.size() > (64u << 20) || !onnx::parse_model(bytes, &m, &e) || !silero_bind(m, w, &e)) { *err = "Failed to load VAD model from '" + path + "': " + (e.empty() ? "file too large" : e);
        return false; }
        return true;
    }
    explicit SileroVad(std::shared_ptr<const SileroWeights> w) : w_(std::move(w)) { reset(); }
    void reset() { memset(h_, 0, sizeof h_); memset(c_, 0, sizeof c_); memset(ctx_, 0, sizeof ctx_); }
    const float* state_h() const { return h_; }
    const float* state_c() const { return c_; }
    float process_chunk(const float* audio512) {
        const SileroWeights& w = *w_;
        float x[640];
        memcpy(x, ctx_, sizeof(float) * 64); memcpy(x + 64, audio512, sizeof(float) * 512);
        for (int j = 0; j < 64; ++j) x[576 + j] = x[574 - j];                 // reflect padding (no edge repeat) of 64 on the right
        float a[129 * 4], b[128 * 4];
        for (int fr = 0; fr < 4; ++fr)
            for (int bin = 0; bin < 129; ++bin) {
                const float* br = &w.basis[(size_t)bin * 256]; const float* bi = &w.basis[(size_t)(129 + bin) * 256]; const float* xs = x + 128 * fr;
                float re = 0.0f, im = 0.0f;
                for (int k = 0; k < 256; ++k) { re += br[k] * xs[k]; im += bi[k] * xs[k]; }
                a[bin * 4 + fr] = sqrtf(re * re + im * im);
            }
        int T = 4; float* in = a; float* out = b;
        for (int l = 0; l < 4; ++l) {
            const int ci = SILERO_CI[l], co = SILERO_CO[l], st = SILERO_STRIDE[l], To = (T + 2 - 3) / st + 1;
            for (int o = 0; o < co; ++o)
                for (int t = 0; t < To; ++t) {
                    float s = w.cb[l][o];
                    for (int c = 0; c < ci; ++c)
                        for (int k = 0; k < 3; ++k) { const int p = t * st - 1 + k; if (p >= 0 && p < T) s += w.cw[l][((size_t)o * ci + c) * 3 + k] * in[c * T + p]; }
                    out[o * To + t] = s > 0.0f ? s : 0.0f;
                }
            T = To; float* tmp = in; in = out; out = tmp;   // ping-pong between the two scratch arrays
        }
        // in: [128][1]
        float gates[512];
        for (int r = 0; r < 512; ++r) {
            float s = w.b_ih[r]; const float* wi = &w.w_ih[(size_t)r * 128]; for (int k = 0; k < 128; ++k) s += wi[k] * in[k];
            float u = w.b_hh[r]; const float* wh = &w.w_hh[(size_t)r * 128]; for (int k = 0; k < 128; ++k) u += wh[k] * h_[k];
            gates[r] = s + u;
        }
        float acc = w.ob;
        for (int j = 0; j < 128; ++j) {
            const float ig = sigm(gates[j]), fg = sigm(gates[128 + j]), gg = tanhf(gates[256 + j]), og = sigm(gates[384 + j]);
            const float c = fg * c_[j] + ig * gg; const float h = og * tanhf(c);
            c_[j] = c; hn_[j] = h;
        }
        memcpy(h_, hn_, sizeof h_);
        for (int j = 0; j < 128; ++j) acc += w.ow[j] * (h_[j] > 0.0f ? h_[j] : 0.0f);
        memcpy(ctx_, audio512 + 512 - 64, sizeof(float) * 64);
        return sigm(acc);
    }
private:
    static float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }
    std::shared_ptr<const SileroWeights> w_;
    float h_[128], c_[128], hn_[128], ctx_[64];
};

}  // namespace skw
