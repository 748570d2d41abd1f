// skw_engine.hip — host side of libskw_engine.so: GGML model loading, device workspace,
// and the batched restatement of whisper_full_with_state (greedy, T = 0) on top of skw_kernels.
//
// Reference call sites this replaces: /root/reference/plugins/native/whisper/src/lib.rs
//   :354-363 model load, :377-379 state, :624-646 params + full(), :650-660 segment readout.
#include "../../include/skw_engine.h"
#include "../../include/skw_math.h"
#include "../../include/skw_ggml_quant.h"
#include "skw_kernels.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define WHISPER_SAMPLE_RATE 16000
#define WHISPER_N_FFT 400
#define WHISPER_HOP 160
#define WHISPER_CHUNK_SIZE 30

static const char* const g_lang[] = {"en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi",
    "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az",
    "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be",
    "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"};
static const int g_n_lang = (int)(sizeof(g_lang) / sizeof(g_lang[0]));
// whisper.cpp's g_lang also maps the full language names; whisper_lang_id accepts either spelling
static const char* const g_lang_name[] = {"english", "chinese", "german", "spanish", "russian", "korean", "french", "japanese", "portuguese", "turkish", "polish", "catalan", "dutch",
    "arabic", "swedish", "italian", "indonesian", "hindi", "finnish", "vietnamese", "hebrew", "ukrainian", "greek", "malay", "czech", "romanian", "danish", "hungarian", "tamil", "norwegian",
    "thai", "urdu", "croatian", "bulgarian", "lithuanian", "latin", "maori", "malayalam", "welsh", "slovak", "telugu", "persian", "latvian", "bengali", "serbian", "azerbaijani", "slovenian",
    "kannada", "estonian", "macedonian", "breton", "basque", "icelandic", "armenian", "nepali", "mongolian", "bosnian", "kazakh", "albanian", "swahili", "galician", "marathi", "punjabi",
    "sinhala", "khmer", "shona", "yoruba", "somali", "afrikaans", "occitan", "georgian", "belarusian", "tajik", "sindhi", "gujarati", "amharic", "yiddish", "lao", "uzbek", "faroese",
    "haitian creole", "pashto", "turkmen", "nynorsk", "maltese", "sanskrit", "luxembourgish", "myanmar", "tibetan", "tagalog", "malagasy", "assamese", "tatar", "hawaiian", "lingala", "hausa",
    "bashkir", "javanese", "sundanese", "cantonese"};
static_assert(sizeof(g_lang_name) / sizeof(g_lang_name[0]) == sizeof(g_lang) / sizeof(g_lang[0]), "one name per code");

static const char* const NST_LIST[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
    "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))",
    "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac",
        "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};
static const int N_NST_LIST = (int)(sizeof(NST_LIST) / sizeof(NST_LIST[0]));

static void set_err(char* err, size_t n, const char* fmt, ...) {
    if (!err || !n) return;
    va_list ap; va_start(ap, fmt); vsnprintf(err, n, fmt, ap); va_end(ap);
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(errbuf, 512, "HIP error '%s' at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

struct DevLin { half_t* w = nullptr; float* b = nullptr; int n_out = 0, n_in = 0, k_pad = 0;
                half_t* w_nat = nullptr;   // the same weight with the contraction axis in natural order (decoder projections fed by a LayerNorm: skw_gemm16_small_lnA)
                half_t* w_frag = nullptr; half_t* w_nat_frag = nullptr;   // fragment-order images of w / w_nat for the f16 decode kernels (SkwGemmArgs::Wf; decoder layers only)
                // ggml's arithmetic for a block-quantised weight (skw_kernels_q8.hip): int8 [n_out][n_in], scales / offsets [n_in / 32][n_pad]; null for f16 weights
                int8_t* qw = nullptr; float* dwT = nullptr; float* mwT = nullptr; int n_pad = 0, qform = 0; };
struct HostQ { std::vector<int8_t> q; std::vector<float> d, m; int n_out = 0, K = 0; };      // a quantised weight in the common integer form, host side
struct DevLN { float* w = nullptr; float* b = nullptr; };
struct EncLayer { DevLN attn_ln, mlp_ln; DevLin q, k, v, o, fc1, fc2; };
struct DecLayer { DevLN attn_ln, cross_ln, mlp_ln; DevLin q, k, v, o, cq, ck, cv, co, fc1, fc2; DevLin qkv; /* q|k|v rows concatenated for the fused decode projection */ };

// qblk: the file's blocks of a quantised tensor (data: its f16 twin)
struct RawT { std::string name; int n_dims = 0; int ne[4] = {1, 1, 1, 1}; int type = 0; std::vector<uint8_t> data; size_t n = 0; std::vector<uint8_t> qblk; int qtype = 0; };

struct skw_model {
    skw_hparams hp{};
    int device = 0;
    std::vector<std::string> tok_str;
    int tok_eot, tok_sot, tok_translate, tok_transcribe, tok_solm, tok_prev, tok_nosp, tok_not, tok_beg;
    int tok_space = -1, tok_sp_dash = -1, tok_sp_quote = -1; std::vector<int> nst_ids; int n_lang = 99;
    // device
    float *filters = nullptr, *hann = nullptr, *sin_t = nullptr, *cos_t = nullptr; int n_fft_bins = 201; int *mel_grp_lo = nullptr, *mel_grp_hi = nullptr;
    uint16_t* gelu_tab = nullptr;
    float* e_pe = nullptr; DevLin conv1, conv2; DevLN ln_post; std::vector<EncLayer> enc;
    float* d_pe = nullptr; DevLin te; DevLN d_ln; std::vector<DecLayer> dec;
    int quant = 0;             // ggml type of the matmul weights when the file is uniformly block-quantised and ggml's q8 arithmetic is available (exact precision runs it)
    float* te32 = nullptr;     // quant: the token embedding dequantised to f32 [n_vocab][d] (get_rows does not round to f16)
    std::vector<void*> allocs;
};

template <typename T> static T* dev_upload(skw_model* m, const T* h, size_t n) {
    T* d = nullptr; if (hipMalloc((void**)&d, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) return nullptr;
    if (n && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    m->allocs.push_back(d); return d;
}
static RawT* find_t(std::vector<RawT>& ts, const std::string& name) { for (auto& t : ts) if (t.name == name) return &t; return nullptr; }
static const float* as_f32(RawT* t, std::vector<float>& tmp) {
    if (t->type == 0) return (const float*)t->data.data();
    tmp.resize(t->n); const uint16_t* h = (const uint16_t*)t->data.data(); for (size_t i = 0; i < t->n; ++i) tmp[i] = skw_f16_to_f32(h[i]); return tmp.data();
}
static bool is_matmul_weight(const RawT& t) {
    const std::string& n = t.name;
    return t.n_dims == 2 && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0 && n.find("positional_embedding") == std::string::npos && n.find("ln") == std::string::npos;
}
static void host_q(const RawT& w, HostQ* h) {
    const int K = w.ne[0], n_out = w.ne[1], nb = K / 32; const size_t bb = skw_ggml_block_bytes(w.qtype);
    h->n_out = n_out; h->K = K; h->q.resize((size_t)n_out * K); h->d.resize((size_t)n_out * nb); h->m.resize((size_t)n_out * nb);
    for (size_t i = 0; i < (size_t)n_out * nb; ++i) skw_ggml_unpack_block(w.qtype, w.qblk.data() + i * bb, h->q.data() + i * 32, &h->d[i], &h->m[i]);
}
// int8 rows as they are; scales and offsets transposed to [block][n_pad] so that a lane's four adjacent features are one 16-byte load
static bool up_q(skw_model* m, const HostQ& h, int qtype, DevLin* L) {
    const int nb = h.K / 32, n_pad = (h.n_out + 63) & ~63;
    std::vector<float> dT((size_t)nb * n_pad, 0.0f), mT((size_t)nb * n_pad, 0.0f);
    for (int n = 0; n < h.n_out; ++n) for (int b = 0; b < nb; ++b) { dT[(size_t)b * n_pad + n] = h.d[(size_t)n * nb + b]; mT[(size_t)b * n_pad + n] = h.m[(size_t)n * nb + b]; }
    L->qw = dev_upload(m, h.q.data(), h.q.size()); L->dwT = dev_upload(m, dT.data(), dT.size()); L->mwT = dev_upload(m, mT.data(), mT.size());
    L->n_pad = n_pad; L->qform = skw_ggml_dot_form(qtype);
    return L->qw && L->dwT && L->mwT;
}
// f16 weight [n_out][n_in] (or conv [oc][ic][kw]) -> device [n_out][k_pad] with the contraction axis in kperm order
static bool up_lin(skw_model* m, std::vector<RawT>& ts, const std::string& wname, const char* bname, DevLin* L, char* err, size_t errlen, bool want_nat = false) {
    RawT* w = find_t(ts, wname);
    if (!w) { set_err(err, errlen, "missing tensor %s", wname.c_str()); return false; }
    if (w->type != 1) { set_err(err, errlen, "tensor %s: only f16 matmul weights are supported (ggml type %d)", wname.c_str(), w->type); return false; }
    int n_in, n_out, kw = 1, ic = 0;
    if (w->n_dims == 2) { n_in = w->ne[0]; n_out = w->ne[1]; }
    else if (w->n_dims == 3) { kw = w->ne[0]; ic = w->ne[1]; n_in = kw * ic; n_out = w->ne[2]; }
    else { set_err(err, errlen, "tensor %s: bad dims", wname.c_str()); return false; }
    const int k_pad = (n_in + 31) & ~31;
    std::vector<uint16_t> h((size_t)n_out * k_pad, 0);
    const uint16_t* src = (const uint16_t*)w->data.data();
    for (int o = 0; o < n_out; ++o) {
        uint16_t* dst = h.data() + (size_t)o * k_pad;
        if (w->n_dims == 2) for (int i = 0; i < n_in; ++i) dst[skw_kperm(i)] = src[(size_t)o * n_in + i];
        else for (int c = 0; c < ic; ++c) for (int t = 0; t < kw; ++t) dst[skw_kperm(t * ic + c)] = src[((size_t)o * ic + c) * kw + t];
    }
    L->n_in = n_in; L->n_out = n_out; L->k_pad = k_pad;
    L->w = (half_t*)dev_upload(m, h.data(), h.size());
    if (!L->w) { set_err(err, errlen, "device allocation failed for %s", wname.c_str()); return false; }
    if (want_nat && w->n_dims == 2 && k_pad == n_in) { L->w_nat = (half_t*)dev_upload(m, src, (size_t)n_out * n_in);
    if (!L->w_nat) { set_err(err, errlen, "device allocation failed for %s", wname.c_str()); return false; } }
    if (m->quant && !w->qblk.empty()) { HostQ h; host_q(*w, &h); if (!up_q(m, h, w->qtype, L)) { set_err(err, errlen, "device allocation failed for %s", wname.c_str()); return false; } }
    L->b = nullptr;
    if (bname) {
        RawT* b = find_t(ts, bname);
        if (!b) { set_err(err, errlen, "missing tensor %s", bname); return false; }
        std::vector<float> tmp; L->b = dev_upload(m, as_f32(b, tmp), b->n);
    }
    return true;
}
static bool up_ln(skw_model* m, std::vector<RawT>& ts, const std::string& wname, const std::string& bname, DevLN* L, char* err, size_t errlen) {
    RawT* w = find_t(ts, wname); RawT* b = find_t(ts, bname);
    if (!w || !b) { set_err(err, errlen, "missing tensor %s / %s", wname.c_str(), bname.c_str()); return false; }
    std::vector<float> t1, t2; L->w = dev_upload(m, as_f32(w, t1), w->n); L->b = dev_upload(m, as_f32(b, t2), b->n); return L->w && L->b;
}

extern "C" int skw_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" void* skw_host_alloc(size_t bytes) { void* p = nullptr;
if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
extern "C" void skw_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int skw_model_lang_id(const char* lang) {
    if (!lang) return -1;
    for (int i = 0; i < g_n_lang; ++i) if (!strcmp(g_lang[i], lang)) return i;
    for (int i = 0; i < g_n_lang; ++i) if (!strcmp(g_lang_name[i], lang)) return i;
    return -1;
}

static skw_model* model_load_impl(const char* path, int device, int quant_mode, char* err, size_t errlen);
extern "C" skw_model* skw_model_load_ex(const char* path, int device, int quant_mode, char* err, size_t errlen) {
    try { return model_load_impl(path, device, quant_mode, err, errlen); }
    catch (const std::exception& e) { set_err(err, errlen, "Failed to load Whisper model from '%s': %s", path ? path : "", e.what()); return nullptr; }   // nothing may unwind across the C ABI
}
extern "C" skw_model* skw_model_load(const char* path, int device, char* err, size_t errlen) { return skw_model_load_ex(path, device, SKW_QUANT_GGML, err, errlen); }
extern "C" int skw_model_quant_type(const skw_model* m) { return m->quant; }
static skw_model* model_load_impl(const char* path, int device, int quant_mode, char* err, size_t errlen) {
    int ndev = skw_device_count();
    if (ndev <= 0) { set_err(err, errlen, "no HIP device available: libskw_engine requires an MI355X (gfx950); there is no CPU fallback"); return nullptr; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "gpu_device %d out of range (%d devices)", device, ndev); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice(%d) failed", device); return nullptr; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, errlen, "Failed to load Whisper model from '%s': cannot open file", path); return nullptr; }
    int32_t magic = 0;
    if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6c) { fclose(f); set_err(err, errlen, "Failed to load Whisper model from '%s': bad magic", path); return nullptr; }
    skw_model* m = new skw_model(); m->device = device;
    auto fail = [&](const char* msg) -> skw_model* { set_err(err, errlen, "Failed to load Whisper model from '%s': %s", path, msg); fclose(f); skw_model_free(m); return nullptr; };
    if (fread(&m->hp, 4, 11, f) != 11) return fail("short hparams");
    {   // a corrupt header must fail here, not as std::bad_alloc across the C ABI
        const skw_hparams& h = m->hp;
        auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
        if (!in(h.n_vocab, 1000, 200000) || !in(h.n_audio_ctx, 1, 4096) || !in(h.n_audio_state, 64, 1536) || !in(h.n_audio_head, 1, 32) || !in(h.n_audio_layer, 1, 64) ||
            !in(h.n_text_ctx, 8, 4096) || !in(h.n_text_state, 64, 1536) || !in(h.n_text_head, 1, 32) || !in(h.n_text_layer, 1, 64) || !in(h.n_mels, 1, 256)) return fail("implausible hparams");
    }
    int32_t nm = 0, nf = 0;
    if (fread(&nm, 4, 1, f) != 1 || fread(&nf, 4, 1, f) != 1) return fail("short mel filter header");
    if (nm < 1 || nm > 256 || nf < 1 || nf > 4096) return fail("implausible mel filter header");
    std::vector<float> filt((size_t)nm * nf);
    if (fread(filt.data(), 4, filt.size(), f) != filt.size()) return fail("short mel filters");
    m->n_fft_bins = nf;
    if (nm != m->hp.n_mels) return fail("mel filter count differs from n_mels");
    int32_t nv = 0; if (fread(&nv, 4, 1, f) != 1) return fail("short vocab");
    if (nv < 0 || nv > 200000) return fail("implausible vocabulary size");
    const int NV = m->hp.n_vocab; m->tok_str.assign(NV, std::string());
    for (int i = 0; i < nv; ++i) {
        uint32_t len = 0; if (fread(&len, 4, 1, f) != 1) return fail("short vocab");
        if (len > 4096) return fail("implausible token length");
        std::string s(len, '\0'); if (len && fread(&s[0], 1, len, f) != len) return fail("short vocab");
        if (i < NV) m->tok_str[i] = s;
    }
    m->tok_eot = 50256; m->tok_sot = 50257; m->tok_translate = 50357; m->tok_transcribe = 50358; m->tok_solm = 50359;
    m->tok_prev = 50360; m->tok_nosp = 50361; m->tok_not = 50362; m->tok_beg = 50363;
    if (NV >= 51865) {
        m->tok_eot++; m->tok_sot++; const int dt = NV - 51865;
        m->tok_translate += 1 + dt; m->tok_transcribe += 1 + dt; m->tok_solm += 1 + dt; m->tok_prev += 1 + dt; m->tok_nosp += 1 + dt; m->tok_not += 1 + dt; m->tok_beg += 1 + dt;
        m->n_lang = 99 + dt;
    }
    for (int i = nv; i < NV; ++i) {
        char buf[64];
        if (i > m->tok_beg) snprintf(buf, sizeof buf, "[_TT_%d]", i - m->tok_beg);
        else if (i == m->tok_eot) snprintf(buf, sizeof buf, "[_EOT_]");
        else if (i == m->tok_sot) snprintf(buf, sizeof buf, "[_SOT_]");
        else if (i == m->tok_translate) snprintf(buf, sizeof buf, "[_TRANSLATE_]");
        else if (i == m->tok_transcribe) snprintf(buf, sizeof buf, "[_TRANSCRIBE_]");
        else if (i == m->tok_solm) snprintf(buf, sizeof buf, "[_SOLM_]");
        else if (i == m->tok_prev) snprintf(buf, sizeof buf, "[_PREV_]");
        else if (i == m->tok_nosp) snprintf(buf, sizeof buf, "[_NOSP_]");
        else if (i == m->tok_not) snprintf(buf, sizeof buf, "[_NOT_]");
        else if (i == m->tok_beg) snprintf(buf, sizeof buf, "[_BEG_]");
        else if (i > m->tok_sot && i <= m->tok_sot + m->n_lang) snprintf(buf, sizeof buf, "[_LANG_%s]", g_lang[std::min(i - m->tok_sot - 1, g_n_lang - 1)]);
        else snprintf(buf, sizeof buf, "[_extra_token_%d]", i);
        m->tok_str[i] = buf;
    }
    { // token_to_id lookups performed by whisper_process_logits (a later duplicate string wins, as std::map::operator[] does)
        std::map<std::string, int> t2i; for (int i = 0; i < NV; ++i) t2i[m->tok_str[i]] = i;
        auto get = [&](const std::string& s) { auto it = t2i.find(s); return it == t2i.end() ? -1 : it->second; };
        m->tok_space = get(" "); m->tok_sp_dash = get(" -"); m->tok_sp_quote = get(" '");
        for (int j = 0; j < N_NST_LIST; ++j) for (int sp = 0; sp < 2; ++sp) { int id = get(std::string(sp ? " " : "") + NST_LIST[j]); if (id >= 0) m->nst_ids.push_back(id); }
    }
    std::vector<RawT> ts;
    for (;;) {
        int32_t nd, len, tt; if (fread(&nd, 4, 1, f) != 1) break;
        if (fread(&len, 4, 1, f) != 1 || fread(&tt, 4, 1, f) != 1) return fail("short tensor header");
        RawT t; t.n_dims = nd; t.type = tt; t.n = 1;
        if (nd < 1 || nd > 4 || len < 0 || len > 255) return fail("corrupt tensor header");
        for (int i = 0; i < nd; ++i) { int32_t e; if (fread(&e, 4, 1, f) != 1) return fail("short tensor header");
        if (e < 1 || e > (1 << 24)) return fail("corrupt tensor header"); t.ne[i] = e; t.n *= (size_t)e; }
        if (t.n > ((size_t)1 << 31)) return fail("corrupt tensor header");
        t.name.resize(len); if (len && fread(&t.name[0], 1, len, f) != (size_t)len) return fail("short tensor name");
        size_t esz = tt == 0 ? 4 : tt == 1 ? 2 : 0;
        if (!esz) {   // block-quantised 2-D weights: decoded to f16 here (include/skw_ggml_quant.h, DEVIATION D4)
            const size_t bb = skw_ggml_block_bytes(tt);
            if (!bb || t.ne[0] % 32) { std::string msg = "tensor " + t.name + ": unsupported ggml type " + std::to_string(tt) + " (f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0 are read)";
            return fail(msg.c_str()); }
            std::vector<uint8_t> blocks(t.n / 32 * bb); if (fread(blocks.data(), 1, blocks.size(), f) != blocks.size()) return fail("short tensor data");
            t.data.resize(t.n * 2); skw_ggml_dequant_to_f16(tt, blocks.data(), t.n, (uint16_t*)t.data.data()); t.type = 1; t.qtype = tt; t.qblk = std::move(blocks);
            ts.push_back(std::move(t)); continue;
        }
        t.data.resize(t.n * esz); if (fread(t.data.data(), 1, t.data.size(), f) != t.data.size()) return fail("short tensor data");
        ts.push_back(std::move(t));
    }
    fclose(f); f = nullptr;
    {   // ggml's q8 arithmetic needs every matmul weight in one block type (what whisper.cpp's quantize writes); anything else runs as the f16 twin
        int qt = 0; bool uniform = quant_mode != 0 && !skw_sw(SW_QUANT_TWIN);
        for (auto& t : ts) if (is_matmul_weight(t)) { if (t.qblk.empty()) uniform = false; else if (!qt) qt = t.qtype; else if (qt != t.qtype) uniform = false; }
        m->quant = (uniform && qt) ? qt : 0;
    }
    auto fail2 = [&]() -> skw_model* { skw_model_free(m); return nullptr; };
    // tables
    m->filters = dev_upload(m, filt.data(), filt.size());
    {   // per mel filter: the range of 4-bin groups with a non-zero tap (the filterbank comes from the model file; Slaney filters are narrow)
        std::vector<int> lo(nm, 0), hi(nm, 0);
        for (int j = 0; j < nm; ++j) { int a = nf, b = -1; for (int k = 0; k < nf; ++k) if (filt[(size_t)j * nf + k] != 0.0f) { a = std::min(a, k);
        b = std::max(b, k); } if (b >= 0) { lo[j] = a / 4; hi[j] = b / 4 + 1; } }
        m->mel_grp_lo = dev_upload(m, lo.data(), lo.size()); m->mel_grp_hi = dev_upload(m, hi.data(), hi.size());
    }
    {
        std::vector<float> sn(WHISPER_N_FFT), cs(WHISPER_N_FFT), hn(WHISPER_N_FFT);
        for (int i = 0; i < WHISPER_N_FFT; ++i) { double theta = (2 * M_PI * i) / WHISPER_N_FFT;
        sn[i] = sinf(theta); cs[i] = cosf(theta); hn[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / (WHISPER_N_FFT))); }
        m->sin_t = dev_upload(m, sn.data(), sn.size()); m->cos_t = dev_upload(m, cs.data(), cs.size()); m->hann = dev_upload(m, hn.data(), hn.size());
        std::vector<uint16_t> gt(65536); for (int i = 0; i < 65536; ++i) gt[i] = skw_gelu_table_entry((uint16_t)i);
        m->gelu_tab = dev_upload(m, gt.data(), gt.size());
    }
    std::vector<float> tmp;
    RawT* t;
    if (!(t = find_t(ts, "encoder.positional_embedding"))) { set_err(err, errlen, "missing encoder.positional_embedding"); return fail2(); }
    m->e_pe = dev_upload(m, as_f32(t, tmp), t->n);
    if (!(t = find_t(ts, "decoder.positional_embedding"))) { set_err(err, errlen, "missing decoder.positional_embedding"); return fail2(); }
    m->d_pe = dev_upload(m, as_f32(t, tmp), t->n);
    bool ok = true;
    // fragment-order images of the decoder projections (SkwGemmArgs::Wf); =0: the f16 decode kernels read weight rows
    const bool wfrag_on = skw_sw(SW_DEC_WFRAG) != 0;
    ok = ok && up_lin(m, ts, "encoder.conv1.weight", "encoder.conv1.bias", &m->conv1, err, errlen);
    // conv1 runs as a product over the im2col image [frames][3 n_mels padded]: the f16 GEMMs step K by 64 (80 bands: 240 -> 256, large-v3's 128: 384)
    if (ok && ((m->conv1.k_pad & 63) || m->conv1.n_in != 3 * m->hp.n_mels || m->hp.n_mels > 128)) {
        set_err(err, errlen, "encoder.conv1.weight: 3 x n_mels = %d taps do not pad to a multiple of 64 (n_mels 80 and 128 are supported)", m->conv1.n_in); ok = false;
    }
    ok = ok && up_lin(m, ts, "encoder.conv2.weight", "encoder.conv2.bias", &m->conv2, err, errlen);
    ok = ok && up_ln(m, ts, "encoder.ln_post.weight", "encoder.ln_post.bias", &m->ln_post, err, errlen);
    m->enc.resize(m->hp.n_audio_layer);
    for (int l = 0; l < m->hp.n_audio_layer && ok; ++l) {
        EncLayer& L = m->enc[l]; std::string p = "encoder.blocks." + std::to_string(l) + ".";
        ok = ok && up_ln(m, ts, p + "attn_ln.weight", p + "attn_ln.bias", &L.attn_ln, err, errlen);
        ok = ok && up_lin(m, ts, p + "attn.query.weight", (p + "attn.query.bias").c_str(), &L.q, err, errlen);
        ok = ok && up_lin(m, ts, p + "attn.key.weight", nullptr, &L.k, err, errlen);
        ok = ok && up_lin(m, ts, p + "attn.value.weight", (p + "attn.value.bias").c_str(), &L.v, err, errlen);
        ok = ok && up_lin(m, ts, p + "attn.out.weight", (p + "attn.out.bias").c_str(), &L.o, err, errlen);
        ok = ok && up_ln(m, ts, p + "mlp_ln.weight", p + "mlp_ln.bias", &L.mlp_ln, err, errlen);
        ok = ok && up_lin(m, ts, p + "mlp.0.weight", (p + "mlp.0.bias").c_str(), &L.fc1, err, errlen);
        ok = ok && up_lin(m, ts, p + "mlp.2.weight", (p + "mlp.2.bias").c_str(), &L.fc2, err, errlen);
    }
    ok = ok && up_lin(m, ts, "decoder.token_embedding.weight", nullptr, &m->te, err, errlen);
    if (ok && m->quant) {   // get_rows on the quantised token embedding: q * d (+ m) in f32, not rounded to f16
        RawT* te = find_t(ts, "decoder.token_embedding.weight"); const size_t bb = skw_ggml_block_bytes(te->qtype); std::vector<float> e32(te->n);
        for (size_t i = 0; i < te->n / 32; ++i) skw_ggml_dequant_block(te->qtype, te->qblk.data() + i * bb, e32.data() + i * 32);
        m->te32 = dev_upload(m, e32.data(), e32.size()); ok = m->te32 != nullptr;
    }
    ok = ok && up_ln(m, ts, "decoder.ln.weight", "decoder.ln.bias", &m->d_ln, err, errlen);
    m->dec.resize(m->hp.n_text_layer);
    for (int l = 0; l < m->hp.n_text_layer && ok; ++l) {
        DecLayer& L = m->dec[l]; std::string p = "decoder.blocks." + std::to_string(l) + ".";
        ok = ok && up_ln(m, ts, p + "attn_ln.weight", p + "attn_ln.bias", &L.attn_ln, err, errlen);
        ok = ok && up_lin(m, ts, p + "attn.query.weight", (p + "attn.query.bias").c_str(), &L.q, err, errlen, true);
        ok = ok && up_lin(m, ts, p + "attn.key.weight", nullptr, &L.k, err, errlen, true);
        ok = ok && up_lin(m, ts, p + "attn.value.weight", (p + "attn.value.bias").c_str(), &L.v, err, errlen, true);
        ok = ok && up_lin(m, ts, p + "attn.out.weight", (p + "attn.out.bias").c_str(), &L.o, err, errlen);
        ok = ok && up_ln(m, ts, p + "cross_attn_ln.weight", p + "cross_attn_ln.bias", &L.cross_ln, err, errlen);
        ok = ok && up_lin(m, ts, p + "cross_attn.query.weight", (p + "cross_attn.query.bias").c_str(), &L.cq, err, errlen, true);
        ok = ok && up_lin(m, ts, p + "cross_attn.key.weight", nullptr, &L.ck, err, errlen);
        ok = ok && up_lin(m, ts, p + "cross_attn.value.weight", (p + "cross_attn.value.bias").c_str(), &L.cv, err, errlen);
        ok = ok && up_lin(m, ts, p + "cross_attn.out.weight", (p + "cross_attn.out.bias").c_str(), &L.co, err, errlen);
        ok = ok && up_ln(m, ts, p + "mlp_ln.weight", p + "mlp_ln.bias", &L.mlp_ln, err, errlen);
        ok = ok && up_lin(m, ts, p + "mlp.0.weight", (p + "mlp.0.bias").c_str(), &L.fc1, err, errlen, true);
        ok = ok && up_lin(m, ts, p + "mlp.2.weight", (p + "mlp.2.bias").c_str(), &L.fc2, err, errlen);
        if (ok) {   // fused q|k|v weight [3d][k_pad] and bias [3d] (k has no bias: zeros; adding 0.0f is exact)
            const int d = L.q.n_out, kp = L.q.k_pad; L.qkv.n_in = L.q.n_in; L.qkv.n_out = 3 * d; L.qkv.k_pad = kp;
            half_t* w = nullptr; float* b = nullptr;
            if (hipMalloc((void**)&w, (size_t)3 * d * kp * 2) != hipSuccess || hipMalloc((void**)&b, (size_t)3 * d * 4) != hipSuccess) { set_err(err, errlen, "device allocation failed"); ok = false; }
            else {
                m->allocs.push_back(w); m->allocs.push_back(b);
                hipMemcpy(w, L.q.w, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                hipMemcpy(w + (size_t)d * kp, L.k.w, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                hipMemcpy(w + (size_t)2 * d * kp, L.v.w, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                hipMemset(b, 0, (size_t)3 * d * 4); hipMemcpy(b, L.q.b, (size_t)d * 4, hipMemcpyDeviceToDevice); hipMemcpy(b + 2 * d, L.v.b, (size_t)d * 4, hipMemcpyDeviceToDevice);
                L.qkv.w = w; L.qkv.b = b;
                if (L.q.w_nat && L.k.w_nat && L.v.w_nat) {      // the concatenation again in natural k order
                    half_t* wn = nullptr;
                    if (hipMalloc((void**)&wn, (size_t)3 * d * kp * 2) == hipSuccess) {
                        m->allocs.push_back(wn);
                        hipMemcpy(wn, L.q.w_nat, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                        hipMemcpy(wn + (size_t)d * kp, L.k.w_nat, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                        hipMemcpy(wn + (size_t)2 * d * kp, L.v.w_nat, (size_t)d * kp * 2, hipMemcpyDeviceToDevice);
                        L.qkv.w_nat = wn;
                    }
                }
            }
            // fragment-order images for the f16 decode kernels (skw_make_wfrag): every decoder projection the small-M kernels multiply by; fc1's rows in its GELU epilogue's order
            if (ok && wfrag_on) {
                auto mk = [&](DevLin& X, int perm) {
                    if ((X.n_out & 15) || (X.k_pad & 31)) return;
                    for (int nat = 0; nat < 2; ++nat) {
                        const half_t* src = nat ? X.w_nat : X.w; if (!src) continue;
                        half_t* img = nullptr; if (hipMalloc((void**)&img, (size_t)X.n_out * X.k_pad * 2) != hipSuccess) continue;      // (no image: the kernels read the rows)
                        m->allocs.push_back(img); skw_make_wfrag(src, X.k_pad, X.n_out, X.k_pad, perm, img, nullptr);
                        (nat ? X.w_nat_frag : X.w_frag) = img;
                    }
                };
                mk(L.qkv, 0); mk(L.o, 0); mk(L.cq, 0); mk(L.co, 0); mk(L.fc1, 1); mk(L.fc2, 0);
                mk(L.ck, 0); mk(L.cv, 0);      // the encoder pass's cross K / V^T products (k_gemm16w streams its weights from these images)
            }
            if (ok && m->quant) {   // the same concatenation in the integer form
                HostQ hq, hk, hv, all; host_q(*find_t(ts, p + "attn.query.weight"), &hq); host_q(*find_t(ts, p + "attn.key.weight"), &hk); host_q(*find_t(ts, p + "attn.value.weight"), &hv);
                all.n_out = 3 * d; all.K = hq.K;
                for (const HostQ* h : {&hq, &hk, &hv}) { all.q.insert(all.q.end(), h->q.begin(), h->q.end());
                all.d.insert(all.d.end(), h->d.begin(), h->d.end()); all.m.insert(all.m.end(), h->m.begin(), h->m.end()); }
                ok = up_q(m, all, m->quant, &L.qkv);
            }
        }
    }
    // Encoder weights as fragment-order images too (k_gemm16w, the f16_mfma precision's big GEMM: weights go from these straight into MFMA operands).  Rows in the
    // order of the epilogue that consumes the product: the kperm'ed-output epilogues (Q / K heads, FC1 and conv1 GELU) want strip row p = logical feature kperm^-1(p).
    if (ok) {
        auto mke = [&](DevLin& X, int perm) {
            if ((X.n_out & 15) || (X.k_pad & 63) || !X.w) return;
            half_t* img = nullptr;
            if (hipMalloc((void**)&img, (size_t)X.n_out * X.k_pad * 2) != hipSuccess) return;      // (no image: k_gemm16 stages the rows through LDS)
            m->allocs.push_back(img);
            skw_make_wfrag(X.w, X.k_pad, X.n_out, X.k_pad, perm, img, nullptr);
            X.w_frag = img;
        };
        mke(m->conv1, 1);
        mke(m->conv2, 0);
        for (EncLayer& L : m->enc) {
            mke(L.q, 1);
            mke(L.k, 1);
            mke(L.v, 0);
            mke(L.o, 0);
            mke(L.fc1, 1);
            mke(L.fc2, 0);
        }
        if (hipDeviceSynchronize() != hipSuccess) ok = false;
    }
    if (!ok) return fail2();
    if ((m->hp.n_audio_state / m->hp.n_audio_head) != 64 || (m->hp.n_text_state / m->hp.n_text_head) != 64 || m->hp.n_audio_state % 128 || m->hp.n_text_state % 128 || m->hp.n_audio_state > 1536) {
        // (a multiple of 128: the decoder's contractions are four contiguous K segments, D3'; every Whisper width — 384, 512, 768, 1024, 1280 — is one)
        set_err(err, errlen, "unsupported model geometry (head dim must be 64, state a multiple of 128 and <= 1536)"); return fail2();
    }
    hipDeviceSynchronize();
    return m;
}
extern "C" void skw_model_free(skw_model* m) { if (!m) return; hipSetDevice(m->device); for (void* p : m->allocs) hipFree(p); delete m; }
extern "C" void skw_model_get_hparams(const skw_model* m, skw_hparams* out) { *out = m->hp; }
extern "C" const char* skw_model_token_text(const skw_model* m, int id, int* len) {
    if (id < 0 || id >= m->hp.n_vocab) { if (len) *len = 0; return ""; }
    if (len) *len = (int)m->tok_str[id].size(); return m->tok_str[id].c_str();
}
extern "C" void skw_full_default_params(skw_full_params* p) {
    memset(p, 0, sizeof *p); p->lang_id = 0; p->suppress_blank = 1; p->suppress_nst = 0; p->max_initial_ts = 1.0f; p->entropy_thold = 2.4f; p->logprob_thold = -1.0f; p->no_speech_thold = 0.6f;
    p->temperature = 0.0f; p->temperature_inc = 0.2f;
}

// ------------------------------------------------------------------ the switchboard (SkwSw, skw_kernels.h)
struct SwDef { const char* name; int def; const char* what; };
static const SwDef g_sw_defs[SW_COUNT] = {
    {"QUANT_TWIN", 0, "1: a block-quantised file runs as its dequantised f16 twin in both precisions (model load)"},
    {"DEC_WFRAG", 1, "0: the decode GEMMs read their weights as rows, no fragment-order images are built (model load)"},
    {"GEMM16W", 1, "0: the encoder / cross-K-V / prompt-pass GEMMs run k_gemm16 (both operands through LDS) instead of k_gemm16w (weights from fragment-order images)"},
    {"GEMM16W_NGROUPS", 0, "k_gemm16w's feature-split tile walk: 0 automatic, 1 off, 2 / 4 / 8 forced"},
    {"XATTN_FRAG", 1, "0: f16_mfma keeps cross K / V^T as rows and runs the two-phase cross attention (context creation)"},
    {"DECODE_GRAPHS", -1, "0: the decode steps are launched eagerly instead of as captured step graphs (context creation)"},
    {"DECODE_GROUPS", 0, "row groups of the decode step, each a step graph on its own stream: 0 = one group (the default), n forced (context creation)"},
    {"DEC_LN_STATS", 1, "0: f16_mfma launches the decode step's LayerNorms instead of normalising inside the consuming GEMM (context creation)"},
    {"PROMPT_PASS", 1, "0: the prompt is fed one token per step instead of in one multi-row pass (context creation)"},
    {"PROMPT_SMALL_GEMM", 0, "1: a prompt pass of >= 256 rows keeps the small-M decode GEMMs instead of the big-tile kernel (f16_mfma)"},
    {"PROMPT_XATTN_MQ", 1, "0: the f16_mfma prompt pass streams cross K / V^T once per prompt token (single-query kernel) instead of once per 128 (multi-query)"},
    {"DEC_AFRAG", 1, "0: the f16_mfma decode step hands attention / FC1 outputs to the next GEMM as rows instead of fragment-order images"},
    {"DEC_ATTN_FASTV", 1, "0: the f16_mfma decode self-attention uses the exact form's per-channel P.V chain"},
    {"Q8_LDS", 1, "0: ggml-arithmetic GEMMs of quantised files take their operands from global memory (the form tails and odd geometries always use)"},
    {"RESAMPLE_SCAN", 0, "1: the linear resampler's index sequence comes from the single-lane sequential walk only"},
    {"RESAMPLE_NO_HOST_WALK", 0, "1: long resampler calls step the chunk starts on the device instead of on the host"},
};
static int g_sw_val[SW_COUNT];
static std::once_flag g_sw_once;
static std::atomic<unsigned> g_sw_epoch{0};
// the one place this library reads its environment: SKW_<NAME> for every row of the table, once
static void sw_init() {
    for (int i = 0; i < SW_COUNT; ++i) { const std::string k = std::string("SKW_") + g_sw_defs[i].name; const char* e = getenv(k.c_str()); g_sw_val[i] = e ? atoi(e) : g_sw_defs[i].def; }
}
int skw_sw(int id) { std::call_once(g_sw_once, sw_init); return g_sw_val[id]; }
unsigned skw_sw_epoch() { return g_sw_epoch.load(); }
extern "C" int skw_debug_switch_count(void) { return SW_COUNT; }
extern "C" const char* skw_debug_switch_name(int i) { return i >= 0 && i < SW_COUNT ? g_sw_defs[i].name : nullptr; }
extern "C" const char* skw_debug_switch_what(int i) { return i >= 0 && i < SW_COUNT ? g_sw_defs[i].what : nullptr; }
extern "C" int skw_debug_switch_default(int i) { return i >= 0 && i < SW_COUNT ? g_sw_defs[i].def : 0; }
static int sw_find(const char* name) { for (int i = 0; i < SW_COUNT; ++i) if (!strcmp(name, g_sw_defs[i].name)) return i; return -1; }
extern "C" int skw_debug_switch_get(const char* name) { const int i = sw_find(name); return i < 0 ? -1 : skw_sw(i); }
// tests flip a switch in-process (takes effect at the point the table says: the next launch, context creation or model load; cached step graphs are re-captured)
extern "C" int skw_debug_switch_set(const char* name, int value) { const int i = sw_find(name); if (i < 0) return -1; (void)skw_sw(i); g_sw_val[i] = value; g_sw_epoch.fetch_add(1); return 0; }

// ------------------------------------------------------------------ context / workspace
struct ProfState;
// tests (SKW_TEST_ALLOC_POISON=1 in tests/conftest.py): workspace buffers of floating type that the engine does not zero are filled with NaNs when a context is created, so that a
// kernel reading what no kernel wrote shows up as a changed result instead of passing on whatever hipMalloc returned (round 5: a resampler flag was read that way)
static int g_alloc_poison = 0; static std::atomic<long> g_alloc_poisoned{0};
extern "C" void skw_debug_alloc_poison(int on) { g_alloc_poison = on; }
extern "C" long skw_debug_alloc_poisoned(void) { return g_alloc_poisoned.load(); }      // buffers filled so far
static void poison_floats(void* p, size_t bytes) { if (g_alloc_poison && p) { (void)hipMemset(p, 0xFF, bytes); g_alloc_poisoned.fetch_add(1); } }
struct skw_ctx {
    int precision = SKW_PRECISION_EXACT;             // SKW_PRECISION_*: which form of the contractions runs (skw_ctx_set_precision)
    int kv_frag_on = 1;                              // f16_mfma: cross K / V^T as fragment-order images (skw_kernels.h, skw_kfrag_off); SKW_XATTN_FRAG=0 keeps the row layouts
    bool kv_frag() const { return kv_frag_on && precision == SKW_PRECISION_F16_MFMA; }
    size_t kclip() const { return (size_t)(kv_frag() ? Tpad : m->hp.n_audio_ctx) * m->hp.n_text_state; }      // cross-K elements per window slot (the buffer is sized for the larger: Tpad rows)
    ProfState* prof = nullptr;                       // per-kernel-class event timing (skw_ctx_profile); per context: contexts run on different host threads
    skw_model* m = nullptr; int max_batch = 0, max_samples = 0, n_len_max = 0, Tpad = 0;
    hipStream_t stream = nullptr; hipEvent_t ev[6] = {};
    hipStream_t cur = nullptr;                       // stream the launch helpers enqueue on (== stream outside the decode groups)
    static const int MAX_GROUPS = 8; hipStream_t gstream[MAX_GROUPS] = {}; hipEvent_t gev[MAX_GROUPS] = {}; int n_groups = 1;
    struct StepGraph { int g, r0, n, precision, ln_stats, kclk; unsigned sw_epoch; SkwLogitParams lp; hipGraphExec_t exec; }; std::vector<StepGraph> step_graphs; int use_graphs = 1;
    char errbuf[512] = {0};
    std::vector<void*> allocs;
    // front end
    float* pcm = nullptr; long* pcm_off = nullptr; int* n_samples = nullptr; int* n_len = nullptr; float* mel = nullptr; float* clip_max = nullptr;
    int *clip_idx = nullptr, *seek = nullptr;
    half_t* im2col = nullptr; half_t* h1 = nullptr;
    // encoder
    float* x = nullptr; half_t* y16 = nullptr; half_t *Qh = nullptr, *Kh = nullptr, *Vt = nullptr; half_t* hbuf = nullptr; float* enc_out32 = nullptr;
    half_t *crossK = nullptr, *crossV = nullptr;
    // decoder
    // ggml q8 arithmetic (quantised files, exact precision): unrounded f32 activations and their q8 blocks
    float* dx = nullptr; half_t *dy16 = nullptr, *dq16 = nullptr, *datt16 = nullptr, *dh16 = nullptr;
    half_t *selfK = nullptr, *selfV = nullptr; float* logits = nullptr;
    float *y32 = nullptr, *h32 = nullptr, *encq32 = nullptr, *dy32 = nullptr, *datt32 = nullptr, *dh32 = nullptr;
    int8_t* q8_a = nullptr; float *q8_d = nullptr, *q8_s = nullptr; int q8_kmax = 0;
    // allocated at the first temperature retry (move_retry_slots)
    half_t *stageK = nullptr, *stageV = nullptr; int* slot_map = nullptr;
    int* prompt_buf = nullptr;                       // [B][SKW_PROMPT_CAP] per-row prompts
    int* row_tok = nullptr;                          // per-row prompt token / detected language scratch
    float* probs = nullptr; uint32_t* rng = nullptr;   // sampled (t > 0) passes: probability workspace, std::mt19937 state per clip
    SkwSeqState* st = nullptr; SkwTokenOut* toks = nullptr; uint8_t* static_mask = nullptr; int static_mask_nst = -1;
    SkwSeqState* h_st = nullptr; SkwTokenOut* h_toks = nullptr; // pinned
    int* h_row_live = nullptr; int* d_row_live = nullptr;      // per-row live flags in pinned host memory and their device-side address: k_dec_sample clears a row's flag itself,
                                                               // so a step ends with no 4-byte copy kernel (4.2 us in the chain of every step) — the host reads the flags after the stream drains
    int max_tok = 0;
    // the prompt ([prev] + past text + sot / language / task) in one multi-row pass instead of one token per step; SKW_PROMPT_PASS=0 or skw_debug_set_prompt_pass(ctx, 0)
    int prompt_pass_on = 1;
    int* pf_meta = nullptr; int pf_nseq = 0, pf_nq_max = 0;      // the pass's sequences on the device: [row0 | nq | slot] x max_batch (the multi-query cross attention of the f16_mfma prompt pass)
    int rows_cap = 0; SkwSeqState* pf_st = nullptr;  // decode-step scratch rows (>= max_batch: the prompt pass runs one row per prompt token) and the prompt pass's per-token pseudo-states
    int ln_stats_on = 1;                             // LayerNorm folded into the decode GEMMs (f16_mfma); SKW_DEC_LN_STATS=0 or skw_debug_set_ln_stats(ctx, 0): LayerNorm kernels
    // profiling: rows of the step about to be launched that are still decoding (finished rows return at once in the attention kernels: their bytes are not booked)
    int live_rows_hint = -1;
    // the in-kernel launch clock of the decode step's cross attention (skw_ctx_kernel_clock): one SkwKClk per (row group, layer); cur_group: the group run_decoder_step is enqueuing
    void* kclk = nullptr; int kclk_on = 0, kclk_khz = 0, cur_group = 0;
    int* forced_dev = nullptr; SkwTraceStep* trace_dev = nullptr;   // [max_batch][max_tok], allocated by the first skw_full_batch_traced
    skw_timing timing{};
    int last_enc_B = 0;
    struct WsEntry { const char* name; void** slot; size_t bytes; bool zero; bool is_float; };      // one workspace buffer: the field it fills and its size (skw_ctx_create)
    std::vector<WsEntry> ws_table;
};
struct SkwKClk;
static SkwKClk* kclk_node(const skw_ctx* c, int g, int l);
static int kclk_reset(skw_ctx* c);
// name of the first workspace buffer whose pointer is null, or nullptr when every entry of the table is allocated
static const char* ws_first_null(const skw_ctx* c) {
    if (c->ws_table.empty()) return "(empty workspace table)";
    for (const skw_ctx::WsEntry& e : c->ws_table) if (!*e.slot) return e.name;
    return nullptr;
}
// every entry point that launches kernels starts here: a context whose table holds a null buffer never reaches a launch
#define WS_READY(c) do { if (const char* nb_ = ws_first_null(c)) { snprintf((c)->errbuf, 512, "workspace buffer '%s' is not allocated", nb_); return -1; } } while (0)
extern "C" const char* skw_ctx_last_error(const skw_ctx* c) { return c->errbuf; }
extern "C" int skw_ctx_set_precision(skw_ctx* c, int precision) {
    if (precision != SKW_PRECISION_EXACT && precision != SKW_PRECISION_F16_MFMA) { snprintf(c->errbuf, 512, "unknown precision %d", precision); return -1; }
    c->precision = precision; return 0;
}
extern "C" int skw_ctx_get_precision(const skw_ctx* c) { return c->precision; }
extern "C" void* skw_ctx_stream(const skw_ctx* c) { return (void*)c->stream; }
extern "C" void skw_ctx_last_timing(const skw_ctx* c, skw_timing* out) { *out = c->timing; }

extern "C" skw_ctx* skw_ctx_create(skw_model* m, int max_batch, int max_samples, char* err, size_t errlen) {
    if (!m || max_batch < 1) { set_err(err, errlen, "bad arguments"); return nullptr; }
    if (hipSetDevice(m->device) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return nullptr; }
    skw_ctx* c = new skw_ctx(); c->m = m; c->max_batch = max_batch;
    if (max_samples <= 0) max_samples = WHISPER_SAMPLE_RATE * 31;
    c->max_samples = max_samples;
    c->n_len_max = (max_samples + WHISPER_SAMPLE_RATE * 30 + 2 * (WHISPER_N_FFT / 2) - WHISPER_N_FFT) / WHISPER_HOP + 1;
    const skw_hparams& hp = m->hp; const int B = max_batch, nc = hp.n_audio_ctx, T = 2 * nc, d = hp.n_audio_state, dt = hp.n_text_state;
    c->Tpad = (nc + 31) & ~31;
    c->kv_frag_on = skw_sw(SW_XATTN_FRAG) != 0;
    bool ok = true;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) ok = false;
    c->cur = c->stream;
    // decode groups: the step kernels are latency-bound chains that leave most CUs idle, so independent row groups run concurrently
    // -1 = default (see skw_full_batch): one row group as a captured step graph in both precisions (dependent chains on two streams do
    // not overlap on this part: tools/probe/probe_stream_overlap.hip)
    c->use_graphs = skw_sw(SW_DECODE_GRAPHS);
    c->ln_stats_on = skw_sw(SW_DEC_LN_STATS) != 0;
    c->prompt_pass_on = skw_sw(SW_PROMPT_PASS) != 0;
    c->rows_cap = std::max(max_batch, std::min(max_batch * (SKW_PROMPT_CAP - 1), 4096));
    c->n_groups = skw_sw(SW_DECODE_GROUPS); if (c->n_groups <= 0) c->n_groups = -1; if (c->n_groups > skw_ctx::MAX_GROUPS) c->n_groups = skw_ctx::MAX_GROUPS;
    for (int g = 0; g < skw_ctx::MAX_GROUPS && g < (c->n_groups < 0 ? 1 : c->n_groups) && ok; ++g) { ok = ok && hipStreamCreateWithFlags(&c->gstream[g], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->gev[g], hipEventDisableTiming) == hipSuccess; }
    for (int i = 0; i < 6 && ok; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    // The workspace, as a table: one buffer per line — name, the context field it fills, element count, zero-filled or not.  Every entry is
    // allocated by the loop below and verified non-null BY NAME (skw_ctx_create fails with the buffer's name; full_batch_impl re-checks the table
    // before its first launch), so a buffer that is declared but never allocated cannot reach a kernel as a null pointer.
    const size_t R = (size_t)c->rows_cap;
    const size_t enc_rows = (size_t)B * nc;
    const int kmax = 4 * std::max(d, dt);
    c->max_tok = hp.n_text_ctx / 2;
    if (m->quant) c->q8_kmax = kmax;
    auto want = [&](const char* name, auto*& field, size_t count, bool zero) {
        c->ws_table.push_back(skw_ctx::WsEntry{name, (void**)&field, count * sizeof(*field), zero, !std::is_integral<std::remove_pointer_t<std::remove_reference_t<decltype(field)>>>::value});
    };
    // front end
    want("pcm", c->pcm, (size_t)B * max_samples, false);
    want("pcm_off", c->pcm_off, B, false);
    want("n_samples", c->n_samples, B, false);
    want("n_len", c->n_len, B, false);
    want("mel", c->mel, (size_t)B * c->n_len_max * hp.n_mels, false);
    want("clip_max", c->clip_max, B, false);
    want("clip_idx", c->clip_idx, B, false);
    want("seek", c->seek, B, false);
    want("row_tok", c->row_tok, B, true);
    want("prompt_buf", c->prompt_buf, (size_t)B * SKW_PROMPT_CAP, true);
    want("im2col", c->im2col, (size_t)B * T * c->m->conv1.k_pad, false);
    want("h1", c->h1, (size_t)B * (T + 2) * d, true);
    // encoder
    want("x", c->x, enc_rows * d, false);
    want("y16", c->y16, enc_rows * d, false);
    want("Qh", c->Qh, (size_t)B * hp.n_audio_head * c->Tpad * 64, true);
    want("Kh", c->Kh, (size_t)B * hp.n_audio_head * c->Tpad * 64, true);
    want("Vt", c->Vt, (size_t)B * hp.n_audio_head * 64 * c->Tpad, true);
    want("hbuf", c->hbuf, enc_rows * 4 * d, false);
    want("enc_out32", c->enc_out32, (size_t)nc * d, false);
    // cross K: zeroed — the fragment-order image's pad keys, rows n_audio_ctx .. Tpad of a slot, are never written
    want("crossK", c->crossK, (size_t)hp.n_text_layer * B * c->Tpad * dt, true);
    // cross V^T per head, keys kperm'ed, pad keys stay zero
    want("crossV", c->crossV, (size_t)hp.n_text_layer * B * hp.n_text_head * 64 * c->Tpad, true);
    // decoder: step scratch rows (sized for the prompt pass), caches, logits, sampler state
    want("dx", c->dx, R * dt, false);
    want("dy16", c->dy16, R * dt, false);
    want("dq16", c->dq16, R * dt, false);
    want("datt16", c->datt16, (R + 16) * dt, false);
    want("dh16", c->dh16, (R + 16) * 4 * dt, false);
    want("pf_st", c->pf_st, R, true);
    want("pf_meta", c->pf_meta, (size_t)3 * B, true);
    want("selfK", c->selfK, (size_t)hp.n_text_layer * B * hp.n_text_ctx * dt, true);
    want("selfV", c->selfV, (size_t)hp.n_text_layer * B * hp.n_text_ctx * dt, true);
    want("logits", c->logits, (size_t)B * hp.n_vocab, false);
    want("st", c->st, B, true);
    want("toks", c->toks, (size_t)B * c->max_tok, true);
    want("probs", c->probs, (size_t)B * skw_probs_row_floats(hp.n_vocab), false);
    want("rng", c->rng, (size_t)B * SKW_RNG_WORDS, true);
    want("static_mask", c->static_mask, skw_static_mask_bytes(hp.n_vocab), true);
    if (m->quant) {      // ggml q8 arithmetic (quantised files, exact precision): unrounded f32 activations and their q8 blocks
        want("y32", c->y32, enc_rows * d, false);
        want("h32", c->h32, enc_rows * 4 * d, false);
        want("encq32", c->encq32, enc_rows * d, false);
        want("dy32", c->dy32, R * dt, false);
        want("datt32", c->datt32, R * dt, false);
        want("dh32", c->dh32, R * 4 * dt, false);
        want("q8_a", c->q8_a, enc_rows * kmax, false);
        want("q8_d", c->q8_d, enc_rows * (kmax / 32), false);
        want("q8_s", c->q8_s, enc_rows * (kmax / 32), false);
    }
    const char* ws_failed = nullptr;
    size_t ws_failed_bytes = 0;
    for (const skw_ctx::WsEntry& e : c->ws_table) {
        if (!ok) break;
        void* p = nullptr;
        if (hipMalloc(&p, e.bytes) != hipSuccess || !p) { ok = false; ws_failed = e.name; ws_failed_bytes = e.bytes; break; }
        c->allocs.push_back(p);
        if (e.zero) (void)hipMemset(p, 0, e.bytes);
        else if (e.is_float) poison_floats(p, e.bytes);       // tests: every f16 / f32 / f64 buffer the engine does not zero starts as NaNs (skw_debug_alloc_poison)
        *e.slot = p;
    }
    if (ok) ws_failed = ws_first_null(c);
    if (ws_failed) {
        set_err(err, errlen, "workspace buffer '%s' could not be allocated (%zu bytes, max_batch %d)", ws_failed, ws_failed_bytes, max_batch);
        skw_ctx_free(c);
        return nullptr;
    }
    ok = ok && hipHostMalloc((void**)&c->h_st, sizeof(SkwSeqState) * B) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_toks, sizeof(SkwTokenOut) * B * c->max_tok) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_row_live, sizeof(int) * B, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer((void**)&c->d_row_live, c->h_row_live, 0) == hipSuccess;
    if (!ok) { set_err(err, errlen, "stream, event or pinned host allocation failed (max_batch %d)", max_batch); skw_ctx_free(c); return nullptr; }
    hipDeviceSynchronize();
    return c;
}
static void prof_free(skw_ctx* c);
extern "C" void skw_ctx_free(skw_ctx* c) {
    if (!c) return; hipSetDevice(c->m->device);
    prof_free(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& sg : c->step_graphs) hipGraphExecDestroy(sg.exec);
    c->step_graphs.clear();
    for (int g = 0; g < skw_ctx::MAX_GROUPS; ++g) { if (c->gstream[g]) { hipStreamSynchronize(c->gstream[g]); hipStreamDestroy(c->gstream[g]); } if (c->gev[g]) hipEventDestroy(c->gev[g]); }
    for (void* p : c->allocs) hipFree(p);
    hipFree(c->stageK); hipFree(c->stageV); hipFree(c->slot_map); hipFree(c->forced_dev); hipFree(c->trace_dev); hipFree(c->kclk);
    if (c->h_st) hipHostFree(c->h_st); if (c->h_toks) hipHostFree(c->h_toks); if (c->h_row_live) hipHostFree(c->h_row_live);
    for (int i = 0; i < 6; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------ debug taps (encoder layer 0, natural layouts; enabled by skw_debug_enable)
static std::map<std::string, std::vector<float>> g_taps; static bool g_taps_on = false;
// host-side view of the fragment-order layouts (tests/test_cpu_layouts.py checks them without a GPU): element offset of the 16-byte chunk that holds (key, feat .. feat + 7) of cross K /
// (feat, positions pos .. pos + 7) of cross V^T
extern "C" long skw_layout_kfrag_off(int slot, int H, int Tpad, int key, int feat) { return skw_kfrag_off(slot, H, Tpad, key, feat); }
extern "C" long skw_layout_vtfrag_off(int slot, int H, int Tpad, int feat, int pos) { return skw_vtfrag_off(slot, H, Tpad, feat, pos); }
extern "C" int skw_layout_kperm(int k) { return skw_kperm(k); }
extern "C" long skw_layout_afrag_off(int m, int p, int K) { return skw_afrag_off(m, p, K); }
// tests: the fragment-order image of a host weight [N][K] (f16 bits), as the decode kernels read it (skw_make_wfrag)
extern "C" int skw_debug_make_wfrag(const uint16_t* w_host, int N, int K, int perm, uint16_t* img_host) {
    if (N <= 0 || K <= 0 || (K & 31)) return -1;
    const size_t n_in = (size_t)N * K, n_out = (size_t)((N + 15) & ~15) * K;
    half_t *dw = nullptr, *di = nullptr;
    if (hipMalloc((void**)&dw, n_in * 2) != hipSuccess || hipMalloc((void**)&di, n_out * 2) != hipSuccess) { hipFree(dw); hipFree(di); return -2; }
    int rc = 0;
    if (hipMemcpy(dw, w_host, n_in * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemset(di, 0xff, n_out * 2) != hipSuccess) rc = -3;
    if (!rc) { skw_make_wfrag(dw, K, N & ~15, K, perm, di, nullptr);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(img_host, di, (size_t)(N & ~15) * K * 2, hipMemcpyDeviceToHost) != hipSuccess) rc = -4; }
    hipFree(dw); hipFree(di); return rc;
}
// tests: f16_mfma cross K / V^T as fragment-order images (one-pass cross attention) / as rows (two-phase kernel); takes effect at the next encoder pass
extern "C" void skw_debug_set_kv_frag(skw_ctx* c, int on) { c->kv_frag_on = on != 0; }
extern "C" void skw_debug_set_prompt_pass(skw_ctx* c, int on) { c->prompt_pass_on = on != 0; }     // tests: the prompt as one pass / one token per step
extern "C" void skw_debug_set_ln_stats(skw_ctx* c, int on) { c->ln_stats_on = on != 0; }      // tests: the decode step with / without the LayerNorm launches (f16_mfma)
extern "C" void skw_debug_enable(int on) { g_taps_on = on != 0; g_taps.clear(); }
extern "C" long skw_debug_get(const char* name, float* out, size_t cap) {
    auto it = g_taps.find(name); if (it == g_taps.end()) return -1; if (out && cap >= it->second.size()) memcpy(out, it->second.data(), it->second.size() * 4); return (long)it->second.size();
}
enum TapLayout { TAP_F32, TAP_F16_KPERM, TAP_HEADS, TAP_VT };
static void tap(skw_ctx* c, const char* name, const void* dev, int rows, int cols, TapLayout lay) {
    if (!g_taps_on) return;
    hipStreamSynchronize(c->stream);
    std::vector<float>& o = g_taps[name]; o.assign((size_t)rows * cols, 0.f);
    const int H = c->m->hp.n_audio_head, Tpad = c->Tpad;
    if (lay == TAP_F32) { hipMemcpy(o.data(), dev, o.size() * 4, hipMemcpyDeviceToHost); return; }
    if (lay == TAP_F16_KPERM) {
        std::vector<uint16_t> h((size_t)rows * cols); hipMemcpy(h.data(), dev, h.size() * 2, hipMemcpyDeviceToHost);
        for (int r = 0; r < rows; ++r) for (int i = 0; i < cols; ++i) o[(size_t)r * cols + i] = skw_f16_to_f32(h[(size_t)r * cols + skw_kperm(i)]);
    } else if (lay == TAP_HEADS) {   // [(h)*Tpad + i][kperm(d)] for batch 0
        std::vector<uint16_t> h((size_t)H * Tpad * 64); hipMemcpy(h.data(), dev, h.size() * 2, hipMemcpyDeviceToHost);
        for (int r = 0; r < rows; ++r) for (int n = 0; n < cols; ++n) o[(size_t)r * cols + n] = skw_f16_to_f32(h[((size_t)(n >> 6) * Tpad + r) * 64 + skw_kperm(n & 63)]);
    } else {                          // V^T: [(h*64 + c)][kperm(key)] row stride Tpad
        std::vector<uint16_t> h((size_t)H * 64 * Tpad); hipMemcpy(h.data(), dev, h.size() * 2, hipMemcpyDeviceToHost);
        for (int r = 0; r < rows; ++r) for (int n = 0; n < cols; ++n) o[(size_t)r * cols + n] = skw_f16_to_f32(h[((size_t)n) * Tpad + skw_kperm(r)]);
    }
}

// ------------------------------------------------------------------ per-kernel-class profiling (HIP events on the engine stream)
enum ProfClass { PC_GEMM = 0, PC_GEMM_SMALL, PC_ATTN_ENC, PC_LAYERNORM, PC_MEL, PC_DEC_ATTN, PC_DEC_SAMPLE, PC_OTHER, PC_DEC_XATTN, PC_COUNT };
static const char* const g_prof_names[PC_COUNT] = {"k_gemm", "k_gemm_smallm", "k_attn_encoder", "k_layernorm", "k_mel", "k_dec_self_attn", "k_dec_sample", "other", "k_dec_cross_attn"};
struct ProfRec { int cls; double flops, bytes; hipEvent_t a, b; };
struct ProfState { bool on = false; std::vector<ProfRec> recs;
std::vector<hipEvent_t> pool; size_t next = 0; double ms[PC_COUNT] = {}, flops[PC_COUNT] = {}, bytes[PC_COUNT] = {}; long count[PC_COUNT] = {}; };
static void prof_free(skw_ctx* c) { if (!c->prof) return; for (hipEvent_t e : c->prof->pool) hipEventDestroy(e); delete c->prof; c->prof = nullptr; }
struct ProfScope {
    ProfState* ps; skw_ctx* c; size_t idx; bool ext;      // ext: the launch stamps the two events itself (hipExtLaunchKernelGGL: kernel begin / end, what rocprofv3 calls the duration)
    ProfScope(skw_ctx* c_, int cls, double flops, double bytes, bool ext_ = false);
    ~ProfScope();
    hipEvent_t ev_a() const; hipEvent_t ev_b() const;
    void cancel() { if (ps) { ps->recs[idx].flops = 0.0; ps->recs[idx].bytes = 0.0; ps->recs[idx].cls = PC_OTHER; } }      // the launch did not happen: book nothing for it
};

// ------------------------------------------------------------------ GEMM helpers
ProfScope::ProfScope(skw_ctx* c_, int cls, double flops, double bytes, bool ext_) : ps(nullptr), c(c_), idx(0), ext(ext_) {
    if (!c_->prof || !c_->prof->on) return;
    ps = c_->prof;
    auto get = [&]() { if (ps->next == ps->pool.size()) { hipEvent_t e; hipEventCreate(&e); ps->pool.push_back(e); } return ps->pool[ps->next++]; };
    ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes; r.a = get(); r.b = get(); idx = ps->recs.size(); ps->recs.push_back(r);
    if (!ext) hipEventRecord(r.a, c->cur);
}
ProfScope::~ProfScope() { if (ps && !ext) hipEventRecord(ps->recs[idx].b, c->cur); }
hipEvent_t ProfScope::ev_a() const { return (ps && ext) ? ps->recs[idx].a : nullptr; }
hipEvent_t ProfScope::ev_b() const { return (ps && ext) ? ps->recs[idx].b : nullptr; }
static void prof_collect(skw_ctx* c) {
    if (!c->prof || !c->prof->on) return; ProfState& ps = *c->prof;
    hipStreamSynchronize(c->stream);
    for (auto& r : ps.recs) { float ms = 0; if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    ps.ms[r.cls] += ms; ps.flops[r.cls] += r.flops; ps.bytes[r.cls] += r.bytes; ps.count[r.cls]++; }
    ps.recs.clear(); ps.next = 0;
}
extern "C" void skw_ctx_profile(skw_ctx* c, int on) { if (!c->prof) c->prof = new ProfState();
ProfState& ps = *c->prof; ps.on = on != 0; for (int i = 0; i < PC_COUNT; ++i) { ps.ms[i] = ps.flops[i] = ps.bytes[i] = 0;
ps.count[i] = 0; } ps.recs.clear(); ps.next = 0; }
extern "C" int skw_ctx_profile_get(skw_ctx* c, int cls, char* name, size_t name_len, long* count, double* ms, double* flops, double* bytes) {
    if (cls < 0 || cls >= PC_COUNT || !c->prof) return -1; ProfState& ps = *c->prof;
    if (name) snprintf(name, name_len, "%s", g_prof_names[cls]); *count = ps.count[cls]; *ms = ps.ms[cls]; *flops = ps.flops[cls]; *bytes = ps.bytes[cls]; return 0;
}
// algorithmic work of one GEMM launch: 2*M*N*K flops; bytes = operands read once + result written once
static void gemm_work(const SkwGemmArgs& a, int k_logical, double* fl, double* by) { *fl = 2.0 * a.M * a.N * k_logical;
*by = 2.0 * ((double)a.M * k_logical + (double)a.N * k_logical) + 2.0 * a.M * a.N; }
static void GEMM(skw_ctx* c, const SkwGemmArgs& a, int k_logical) {
    double fl, by; gemm_work(a, k_logical, &fl, &by); ProfScope p(c, PC_GEMM, fl, by);
    if (c->precision == SKW_PRECISION_F16_MFMA && (a.K & 63) == 0) skw_gemm16(a, c->cur);   // K step of the f16 kernel is 64; every Whisper geometry satisfies it
    else skw_gemm(a, c->cur);
}
static void GEMM_S(skw_ctx* c, const SkwGemmArgs& a, int k_logical) {
    double fl, by; gemm_work(a, k_logical, &fl, &by); ProfScope p(c, PC_GEMM_SMALL, fl, by);
    // (a VALU row-parallel variant — lane = batch row, v_fma_mix chains — was measured slower: a dependent v_fma costs 8-10 cycles
    //  on gfx950, no better than the MFMA's 40 cycles per 4 k)
    if (c->precision == SKW_PRECISION_F16_MFMA && skw_gemm16_small(a, c->cur)) return;
    skw_gemm_smallm(a, c->cur);
}

// decode GEMM fed by a LayerNorm of x, as two launches (the exact precision; f16_mfma where the normalising GEMM does not cover the geometry)
static void GEMM_S(skw_ctx* c, const SkwGemmArgs& a, int k_logical);
static void GEMM_LN(skw_ctx* c, SkwGemmArgs a, const float* x, const DevLN& ln, half_t* y16, hipStream_t s, bool normalised = false) {
    if (normalised) { GEMM_S(c, a, a.K); return; }                     // y16 already holds LayerNorm(x): the embedding kernel did it (layer 0)
    { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * a.M * a.K); skw_layernorm(x, a.M, a.K, ln.w, ln.b, y16, nullptr, s); }
    GEMM_S(c, a, a.K);
}

// ---- ggml's arithmetic for quantised files (skw_kernels_q8.hip): quantise the f32 rows once, then any number of products with them
static bool use_q8(const skw_ctx* c) { return c->m->quant != 0 && c->precision == SKW_PRECISION_EXACT; }
static void Q8_ROWS(skw_ctx* c, const float* act32, long lda, int M, int K, int r0) {
    ProfScope p(c, PC_LAYERNORM, 0, 5.0 * M * K);
    skw_q8_quantize(act32, lda, M, K, c->q8_a + (size_t)r0 * c->q8_kmax, c->q8_d + (size_t)r0 * (c->q8_kmax / 32), c->q8_s + (size_t)r0 * (c->q8_kmax / 32), c->cur);
}
// LayerNorm -> f32 rows -> q8 blocks.  (A LayerNorm kernel that quantises its own rows was measured slower than the two launches, twice: one wave
// then owns a whole row's 24 blocks — 266 - 298 vs 243 ms of decode per batch — where the quantiser has a thread per block.)
static void Q8_LN(skw_ctx* c, const float* x, int M, int d, const DevLN& ln, int r0, float* y32) {
    { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * M * d); skw_layernorm(x, M, d, ln.w, ln.b, nullptr, y32, c->cur); }
    Q8_ROWS(c, y32, d, M, d, r0);
}
static void Q8_GEMM(skw_ctx* c, SkwGemmArgs a, const DevLin& L, int r0, bool decoder = false) {
    a.K = L.n_in; a.N = L.n_out; a.bias = L.b;
    SkwQ8Args qa{c->q8_a + (size_t)r0 * c->q8_kmax, c->q8_d + (size_t)r0 * (c->q8_kmax / 32), c->q8_s + (size_t)r0 * (c->q8_kmax / 32), L.qw, L.dwT, L.mwT, L.n_pad, L.qform,
                 decoder && !(L.n_in & 127) ? 1 : 0};
    ProfScope p(c, a.M <= 64 ? PC_GEMM_SMALL : PC_GEMM, 2.0 * a.M * a.N * a.K, 1.0 * a.M * a.K + 1.0 * a.N * a.K + 4.0 * a.M * a.N);
    skw_gemm_q8(a, qa, c->cur);
}
static SkwGemmArgs q8_args(int M, void* C, long ldc, int epi) { SkwGemmArgs a{}; a.M = M; a.C = C; a.ldc = ldc; a.epi = epi; a.scale = 1.0f; return a; }

static SkwGemmArgs gemm_args(const half_t* A, long lda, const DevLin& L, int M, void* C, long ldc, int epi) {
    SkwGemmArgs a{}; a.A = A; a.lda = lda; a.W = L.w; a.Wf = L.w_frag; a.ldw = L.k_pad; a.M = M; a.N = L.n_out; a.K = L.k_pad; a.C = C;
    a.ldc = ldc; a.bias = L.b; a.epi = epi; a.scale = 1.0f; return a;
}

// front end for `n` clips already described in c->pcm_off / n_samples / n_len (device): mel + normalisation
static void run_mel(skw_ctx* c, int n) {
    skw_model* m = c->m; SkwMelTables t{m->hann, m->sin_t, m->cos_t, m->filters, m->hp.n_mels, m->n_fft_bins, m->mel_grp_lo, m->mel_grp_hi};
    ProfScope p_(c, PC_MEL, 0, 0);
    skw_mel_frames(c->pcm, c->pcm_off, c->n_samples, c->n_len, n, c->n_len_max, t, c->mel, c->stream);
    skw_mel_normalize(c->mel, c->n_len, n, c->n_len_max, m->hp.n_mels, c->clip_max, c->stream);
}
// conv stem for Bw window slots (clip_idx/seek on device) -> c->x [Bw*nc][d]
// row0: the first window slot to compute (slots before it keep what they hold: temperature retries, see skw_full_batch); the stem's
// buffers are scratch, so the Bw - row0 computed windows sit at their start
static void run_conv(skw_ctx* c, int Bw_all, int row0 = 0) {
    skw_model* m = c->m; const int nc = m->hp.n_audio_ctx, T = 2 * nc, d = m->hp.n_audio_state; const int Bw = Bw_all - row0;
    skw_mel_im2col(c->mel, c->clip_idx + row0, c->seek + row0, c->n_len, Bw, c->n_len_max, m->hp.n_mels, T, m->conv1.k_pad, c->im2col, c->stream);
    SkwGemmArgs a = gemm_args(c->im2col, m->conv1.k_pad, m->conv1, Bw * T, c->h1, d, EPI_GELU_F16_KPERM_ROWPAD); a.gelu_tab = m->gelu_tab; a.n_ctx = T;
    GEMM(c, a, m->conv1.n_in);
    SkwGemmArgs b = gemm_args(c->h1, 2L * d, m->conv2, Bw * nc, c->x, d, EPI_CONV2); b.a_rows_per_batch = nc; b.a_batch_stride = (long)(T + 2) * d;
    b.gelu_tab = m->gelu_tab; b.pe = m->e_pe; b.n_ctx = nc;
    GEMM(c, b, m->conv2.n_in);
}
// encoder blocks + ln_post (+ cross K/V) over Bw windows; input c->x
static void run_encoder(skw_ctx* c, int Bw_all, bool want_f32_out, bool cross, int row0 = 0) {
    skw_model* m = c->m; const skw_hparams& hp = m->hp; const int nc = hp.n_audio_ctx, d = hp.n_audio_state, H = hp.n_audio_head; const int Bw = Bw_all - row0, M = Bw * nc;
    const size_t xk0 = (size_t)row0 * c->kclip(), xv0 = (size_t)row0 * hp.n_text_head * 64 * c->Tpad;      // cross K / V of the computed windows land in slots row0 ..
    if (use_q8(c)) {
        // Quantised file, exact precision: every weight product is ggml's (rows -> q8 blocks, integer block dots), so what feeds a
        // projection stays f32 and unrounded — LayerNorm, attention and GELU write f32 here — and only the attention operands
        // (Q, K, V^T: ggml casts those to f16 itself) are f16 as before.  The conv stem's kernels are 3-D tensors: never quantised.
        for (int l = 0; l < hp.n_audio_layer; ++l) {
            const EncLayer& L = m->enc[l];
            Q8_LN(c, c->x, M, d, L.attn_ln, 0, c->y32);
            { SkwGemmArgs a = q8_args(M, c->Qh, 0, EPI_HEADS_F16); a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; Q8_GEMM(c, a, L.q, 0); }
            { SkwGemmArgs a = q8_args(M, c->Kh, 0, EPI_HEADS_F16); a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; Q8_GEMM(c, a, L.k, 0); }
            { SkwGemmArgs a = q8_args(M, c->Vt, 0, EPI_VT_F16); a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; Q8_GEMM(c, a, L.v, 0); }
            { ProfScope p_(c, PC_ATTN_ENC, 4.0 * Bw * H * (double)nc * nc * 64, 2.0 * 4 * M * d);
              skw_attn_encoder(c->Qh, c->Kh, c->Vt, (half_t*)c->y32, d, Bw, H, nc, c->Tpad, c->stream, nullptr, nullptr, 1); }
            Q8_ROWS(c, c->y32, d, M, d, 0);
            { SkwGemmArgs a = q8_args(M, c->x, d, EPI_F32); a.res = c->x; a.ldres = d; Q8_GEMM(c, a, L.o, 0); }
            Q8_LN(c, c->x, M, d, L.mlp_ln, 0, c->y32);
            { SkwGemmArgs a = q8_args(M, c->h32, 4L * d, EPI_GELU_F32); a.gelu_tab = m->gelu_tab; Q8_GEMM(c, a, L.fc1, 0); }
            Q8_ROWS(c, c->h32, 4L * d, M, 4 * d, 0);
            { SkwGemmArgs a = q8_args(M, c->x, d, EPI_F32); a.res = c->x; a.ldres = d; Q8_GEMM(c, a, L.fc2, 0); }
        }
        skw_layernorm(c->x, M, d, m->ln_post.w, m->ln_post.b, nullptr, c->encq32, c->stream);
        if (want_f32_out) hipMemcpyAsync(c->enc_out32, c->encq32, sizeof(float) * (size_t)nc * d, hipMemcpyDeviceToDevice, c->stream);
        if (cross) {
            const int dt = hp.n_text_state; const float Kscale = (float)pow((double)((float)dt / hp.n_text_head), -0.25);
            Q8_ROWS(c, c->encq32, d, M, d, 0);
            for (int l = 0; l < hp.n_text_layer; ++l) {
                const DecLayer& L = m->dec[l];
                half_t* ck = c->crossK + (size_t)l * c->max_batch * c->kclip() + xk0; half_t* cv = c->crossV + (size_t)l * c->max_batch * hp.n_text_head * 64 * c->Tpad + xv0;
                { SkwGemmArgs a = q8_args(M, ck, dt, EPI_F16_PLAIN); a.scale = Kscale; a.has_scale = 1; Q8_GEMM(c, a, L.ck, 0); }
                { SkwGemmArgs a = q8_args(M, cv, 0, EPI_VT_F16); a.n_ctx = nc; a.H = hp.n_text_head; a.Tpad = c->Tpad; Q8_GEMM(c, a, L.cv, 0); }
            }
        }
        c->last_enc_B = Bw_all;
        return;
    }
    for (int l = 0; l < hp.n_audio_layer; ++l) {
        const EncLayer& L = m->enc[l];
        { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * M * d); skw_layernorm(c->x, M, d, L.attn_ln.w, L.attn_ln.b, c->y16, nullptr, c->stream); }
        if (l == 0) tap(c, "l0.ln1", c->y16, nc, d, TAP_F16_KPERM);
        { SkwGemmArgs a = gemm_args(c->y16, d, L.q, M, c->Qh, 0, EPI_HEADS_F16); a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; GEMM(c, a, d); }
        { SkwGemmArgs a = gemm_args(c->y16, d, L.k, M, c->Kh, 0, EPI_HEADS_F16); a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; GEMM(c, a, d); }
        { // V^T via the swapped product: rows = features (weights as the A operand), columns = tokens
            SkwGemmArgs a{}; a.A = L.v.w; a.lda = L.v.k_pad; a.W = c->y16; a.ldw = d; a.M = L.v.n_out; a.N = M; a.K = L.v.k_pad; a.C = c->Vt;
            a.bias = L.v.b; a.epi = EPI_VT_F16; a.n_ctx = nc; a.H = H; a.Tpad = c->Tpad; a.scale = 1.0f;
            // the f16 kernel takes V^T in the natural orientation (tokens x features)
            if (c->precision == SKW_PRECISION_F16_MFMA) { a.A = c->y16; a.lda = d; a.W = L.v.w; a.Wf = L.v.w_frag; a.ldw = L.v.k_pad; a.M = M; a.N = L.v.n_out; }
            GEMM(c, a, d);
        }
        if (l == 0) { tap(c, "l0.q", c->Qh, nc, d, TAP_HEADS); tap(c, "l0.k", c->Kh, nc, d, TAP_HEADS); tap(c, "l0.v", c->Vt, nc, d, TAP_VT); }
        float* dbg = nullptr;
        if (l == 0 && g_taps_on) { hipMalloc((void**)&dbg, sizeof(float) * ((size_t)nc * d + 2 * (size_t)H * nc + 64)); hipMemset(dbg, 0, sizeof(float) * ((size_t)nc * d + 2 * (size_t)H * nc)); }
        float* dbg2 = nullptr;
        if (dbg) { hipMalloc((void**)&dbg2, sizeof(float) * 64 * c->Tpad); hipMemset(dbg2, 0, sizeof(float) * 64 * c->Tpad); }
        { ProfScope p_(c, PC_ATTN_ENC, 4.0 * Bw * H * (double)nc * nc * 64, 2.0 * 4 * M * d);
          if (c->precision == SKW_PRECISION_F16_MFMA && !dbg) skw_attn_encoder16(c->Qh, c->Kh, c->Vt, c->y16, d, Bw, H, nc, c->Tpad, c->stream);
          else skw_attn_encoder(c->Qh, c->Kh, c->Vt, c->y16, d, Bw, H, nc, c->Tpad, c->stream, dbg, dbg2); }
        if (dbg2) { tap(c, "l0.SP", dbg2, 64, c->Tpad, TAP_F32); hipFree(dbg2); }
        if (dbg) { tap(c, "l0.att32", dbg, nc, d, TAP_F32); tap(c, "l0.rmax", dbg + (size_t)nc * d, H, nc, TAP_F32);
        tap(c, "l0.rinv", dbg + (size_t)nc * d + (size_t)H * nc, H, nc, TAP_F32); hipFree(dbg); }
        if (l == 0) tap(c, "l0.att", c->y16, nc, d, TAP_F16_KPERM);
        { SkwGemmArgs a = gemm_args(c->y16, d, L.o, M, c->x, d, EPI_F32); a.res = c->x; a.ldres = d; GEMM(c, a, d); }
        if (l == 0) tap(c, "l0.x1", c->x, nc, d, TAP_F32);
        { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * M * d); skw_layernorm(c->x, M, d, L.mlp_ln.w, L.mlp_ln.b, c->y16, nullptr, c->stream); }
        if (l == 0) tap(c, "l0.ln2", c->y16, nc, d, TAP_F16_KPERM);
        { SkwGemmArgs a = gemm_args(c->y16, d, L.fc1, M, c->hbuf, 4L * d, EPI_GELU_F16_KPERM); a.gelu_tab = m->gelu_tab; GEMM(c, a, d); }
        if (l == 0) tap(c, "l0.h", c->hbuf, nc, 4 * d, TAP_F16_KPERM);
        { SkwGemmArgs a = gemm_args(c->hbuf, 4L * d, L.fc2, M, c->x, d, EPI_F32); a.res = c->x; a.ldres = d; GEMM(c, a, 4 * d); }
        if (l == 0) tap(c, "l0.x2", c->x, nc, d, TAP_F32);
    }
    skw_layernorm(c->x, M, d, m->ln_post.w, m->ln_post.b, c->y16, want_f32_out ? c->enc_out32 : nullptr, c->stream);
    if (cross) {
        const int dt = hp.n_text_state; const float Kscale = (float)pow((double)((float)dt / hp.n_text_head), -0.25);
        for (int l = 0; l < hp.n_text_layer; ++l) {
            const DecLayer& L = m->dec[l];
            half_t* ck = c->crossK + (size_t)l * c->max_batch * c->kclip() + xk0; half_t* cv = c->crossV + (size_t)l * c->max_batch * hp.n_text_head * 64 * c->Tpad + xv0;
            { SkwGemmArgs a = gemm_args(c->y16, d, L.ck, M, ck, dt, EPI_F16_PLAIN); a.scale = Kscale; a.has_scale = 1; if (c->kv_frag()) { a.frag = 1;
            a.n_ctx = nc; a.H = hp.n_text_head; a.Tpad = c->Tpad; } GEMM(c, a, d); }
            { // cross V^T through the operand-swapped product (rows = features, columns = tokens), as for the encoder's V
                SkwGemmArgs a{}; a.A = L.cv.w; a.lda = L.cv.k_pad; a.W = c->y16; a.ldw = d; a.M = L.cv.n_out; a.N = M; a.K = L.cv.k_pad; a.C = cv;
                a.bias = L.cv.b; a.epi = EPI_VT_F16; a.n_ctx = nc; a.H = hp.n_text_head; a.Tpad = c->Tpad; a.scale = 1.0f;
                if (c->precision == SKW_PRECISION_F16_MFMA) { a.A = c->y16; a.lda = d; a.W = L.cv.w; a.ldw = L.cv.k_pad; a.M = M; a.N = L.cv.n_out; a.Wf = L.cv.w_frag; a.frag = c->kv_frag(); }
                GEMM(c, a, d);
            }
        }
    }
    c->last_enc_B = Bw_all;
}

// one decoder step for Bw sequences: token/pos taken from the device state; logits computed when want_logits
__global__ void k_set_tokens(SkwSeqState* st, int tok, int pos) { st[blockIdx.x].cur_token = tok; st[blockIdx.x].cur_pos = pos; st[blockIdx.x].active = 1; }
// whisper_lang_auto_detect_with_state: the language whose token has the largest logit after the [sot] step (lowest id on a tie)
__global__ void k_lang_argmax(const float* logits, int n_vocab, int tok_sot, int n_lang, int* out) {
    const float* lg = logits + (long)blockIdx.x * n_vocab; float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n_lang; i += 64) { float v = lg[tok_sot + 1 + i]; if (v > bv) { bv = v; bi = i; } }
    for (int o = 32; o > 0; o >>= 1) { float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64); if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
    if (threadIdx.x == 0) out[blockIdx.x] = bi;
}
// rows [r0, r0 + Bw) of the window batch on stream s: sequences are independent, so groups of rows can run on their own streams
// prefill: the rows are prompt tokens (c->pf_st, scratch from row 0), several per sequence: the K / V caches are addressed through the row's sequence (pad) and cache row (seek)
static void run_decoder_step(skw_ctx* c, int r0, int Bw, int pos, bool want_logits, hipStream_t s, bool prefill = false) {
    skw_model* m = c->m; const skw_hparams& hp = m->hp; const int dt = hp.n_text_state, H = hp.n_text_head, nc = hp.n_audio_ctx, ntc = hp.n_text_ctx;
    const float KQscale = (float)pow((double)((float)dt / H), -0.25);
    const int live = c->live_rows_hint >= 0 ? std::min(c->live_rows_hint, Bw) : Bw;      // rows whose attention kernels do work (algorithmic-byte booking of the profile)
    c->cur = s;
    float* dx = c->dx + (size_t)r0 * dt; half_t* dy16 = c->dy16 + (size_t)r0 * dt; half_t* dq16 = c->dq16 + (size_t)r0 * dt; half_t* datt16 = c->datt16 + (size_t)r0 * dt;
    half_t* dh16 = c->dh16 + (size_t)r0 * 4 * dt; SkwSeqState* st = prefill ? c->pf_st : c->st + r0;
    const int* seqp = prefill ? &st[0].pad : nullptr;                                  // row -> sequence for the attention kernels
    // where the QKV epilogue appends a row's K / V: absolute cache row (prefill) or position inside the row's own cache
    const int* kvpos = prefill ? &st[0].seek : &st[0].cur_pos;
    const long kv_ld = prefill ? 0 : (long)ntc * dt;
    if (use_q8(c)) {   // quantised file, exact precision: ggml's arithmetic (see run_encoder); the row group's q8 scratch starts at its first row
        float* dy32 = c->dy32 + (size_t)r0 * dt; float* datt32 = c->datt32 + (size_t)r0 * dt; float* dh32 = c->dh32 + (size_t)r0 * 4 * dt;
        skw_dec_embed_f32(m->te32, m->d_pe, &st[0].cur_token, &st[0].cur_pos, Bw, dt, dx, s);
        for (int l = 0; l < hp.n_text_layer; ++l) {
            const DecLayer& L = m->dec[l];
            half_t* sk = c->selfK + ((size_t)l * c->max_batch + r0) * ntc * dt; half_t* sv = c->selfV + ((size_t)l * c->max_batch + r0) * ntc * dt;
            half_t* ck = c->crossK + ((size_t)l * c->max_batch + r0) * c->kclip(); half_t* cv = c->crossV + ((size_t)l * c->max_batch + r0) * H * 64 * c->Tpad;
            Q8_LN(c, dx, Bw, dt, L.attn_ln, r0, dy32);
            { SkwGemmArgs a = q8_args(Bw, dq16, dt, EPI_DEC_QKV); a.scale = KQscale; a.has_scale = 1; a.n_ctx = dt; a.C2 = sk; a.C3 = sv; a.ldc2 = kv_ld;
            a.pos_ptr = kvpos; a.pos_stride = (int)(sizeof(SkwSeqState) / sizeof(int)); Q8_GEMM(c, a, L.qkv, r0, true); }
            { ProfScope p_(c, PC_DEC_ATTN, 0, 4.0 * live * (pos + 1) * dt);
              SkwQ8Out qo{c->q8_a + (size_t)r0 * c->q8_kmax, c->q8_d + (size_t)r0 * (c->q8_kmax / 32), c->q8_s + (size_t)r0 * (c->q8_kmax / 32), Bw};
              skw_dec_self_attn(dq16, sk, sv, &st[0].cur_pos, Bw, H, dt, ntc, nullptr, &st[0].active, s, 0, qo, seqp); }
            { SkwGemmArgs a = q8_args(Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; Q8_GEMM(c, a, L.o, r0, true); }
            Q8_LN(c, dx, Bw, dt, L.cross_ln, r0, dy32);
            { SkwGemmArgs a = q8_args(Bw, dq16, dt, EPI_F16_PLAIN); a.scale = KQscale; a.has_scale = 1; Q8_GEMM(c, a, L.cq, r0, true); }
            { ProfScope p_(c, PC_DEC_XATTN, 4.0 * live * (double)nc * dt, 4.0 * live * (double)nc * dt);
            skw_dec_cross_attn_vt(dq16, ck, cv, Bw, H, dt, nc, c->Tpad, (half_t*)datt32, &st[0].active, s, 1, 0, seqp); }
            Q8_ROWS(c, datt32, dt, Bw, dt, r0);
            { SkwGemmArgs a = q8_args(Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; Q8_GEMM(c, a, L.co, r0, true); }
            Q8_LN(c, dx, Bw, dt, L.mlp_ln, r0, dy32);
            { SkwGemmArgs a = q8_args(Bw, dh32, 4L * dt, EPI_GELU_F32); a.gelu_tab = m->gelu_tab; Q8_GEMM(c, a, L.fc1, r0, true); }
            Q8_ROWS(c, dh32, 4L * dt, Bw, 4 * dt, r0);
            { SkwGemmArgs a = q8_args(Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; Q8_GEMM(c, a, L.fc2, r0, true); }
        }
        if (want_logits) {
            Q8_LN(c, dx, Bw, dt, m->d_ln, r0, dy32);
            SkwGemmArgs a = q8_args(Bw, c->logits + (size_t)r0 * hp.n_vocab, hp.n_vocab, EPI_F32); Q8_GEMM(c, a, m->te, r0, true);
        }
        c->cur = c->stream;
        return;
    }
    // The first layer's LayerNorm rides on the embedding kernel (same bits: skw_ln_rows).  (The others riding on the GEMM that completes x — the last workgroup to arrive per
    // 16-row block normalises it — measured 15.6 us for GEMM + tail against 5.1 + 5.0 for the two launches, profiles/r02c: a launch boundary is cheaper on this part than a
    // hand-over inside a kernel.  Removed in round 5.)
    const bool embed_ln = dt <= 1536;
    // LayerNorm without a launch (f16_mfma): the GEMM that consumes LayerNorm(x) loads the f32 rows, takes their statistics from its own
    // registers and normalises on the way into the MFMA.  35 of the step's 136 launches go.  SKW_DEC_LN_STATS=0 restores the LayerNorm kernels.
    const bool lnA = c->ln_stats_on && embed_ln && c->precision == SKW_PRECISION_F16_MFMA && (dt & 127) == 0 && m->dec[0].qkv.w_nat && m->dec[0].cq.w_nat
        && m->dec[0].fc1.w_nat && m->dec[0].cq.k_pad == dt;
    // the prompt pass of a long-form batch is thousands of rows: there the projections are the encoder's big-tile GEMM (f16_mfma; the small-M kernels stream the
    // weights once per 16 rows and reach ~50 TF/s at M = 4096, the big kernel 600).  The QKV product keeps the decode form: its epilogue appends to the K / V caches.
    const bool bigM = prefill && c->precision == SKW_PRECISION_F16_MFMA && Bw >= 256 && !skw_sw(SW_PROMPT_SMALL_GEMM);
    const bool afrag_on = skw_sw(SW_DEC_AFRAG) != 0;
    // the attention kernels leave their rows as the fragment-order A image the out-projections read (f16_mfma, small-M kernels on both sides; the cross attention: the one-pass
    //  kernel / its multi-query prompt form)
    const bool sa_frag = afrag_on && c->precision == SKW_PRECISION_F16_MFMA && !bigM && (dt & 127) == 0, xa_frag = sa_frag && c->kv_frag();
    auto gemm_s = [&](const SkwGemmArgs& a) { if (bigM) GEMM(c, a, a.K); else GEMM_S(c, a, a.K); };
    // a GEMM fed by LayerNorm(dx): the normalising form (site >= 0), else LayerNorm kernel + GEMM
    auto gemm_ln = [&](SkwGemmArgs a, const DevLin& Lw, const DevLN& ln, int site, bool normalised) {
        if (bigM && a.epi != EPI_DEC_QKV && !normalised) {
            { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * a.M * a.K); skw_layernorm(dx, a.M, a.K, ln.w, ln.b, dy16, nullptr, s); }
            GEMM(c, a, a.K); return;
        }
        if (lnA && site >= 0) {
            a.W = Lw.w_nat; a.Wf = Lw.w_nat_frag; a.ln_x = dx; a.ln_w = ln.w; a.ln_b = ln.b;
            {   // (the scope ends before the fall-back below books the same product a second time: ADVICE r3)
                ProfScope p(c, PC_GEMM_SMALL, 2.0 * a.M * a.N * a.K, 4.0 * a.M * a.K + 2.0 * a.N * a.K + 2.0 * a.M * a.N);
                if (skw_gemm16_small_lnA(a, c->cur)) return;
                p.cancel();
            }
            a.W = Lw.w; a.Wf = Lw.w_frag; a.ln_x = nullptr;
        }
        GEMM_LN(c, a, dx, ln, dy16, s, normalised);
    };
    if (embed_ln) skw_dec_embed_ln(m->te.w, m->d_pe, &st[0].cur_token, &st[0].cur_pos, Bw, dt, dx, m->dec[0].attn_ln.w, m->dec[0].attn_ln.b, dy16, s);
    else skw_dec_embed(m->te.w, m->d_pe, &st[0].cur_token, &st[0].cur_pos, Bw, dt, dx, s);
    for (int l = 0; l < hp.n_text_layer; ++l) {
        const DecLayer& L = m->dec[l];
        half_t* sk = c->selfK + ((size_t)l * c->max_batch + r0) * ntc * dt; half_t* sv = c->selfV + ((size_t)l * c->max_batch + r0) * ntc * dt;
        half_t* ck = c->crossK + ((size_t)l * c->max_batch + r0) * c->kclip(); half_t* cv = c->crossV + ((size_t)l * c->max_batch + r0) * H * 64 * c->Tpad;
        { SkwGemmArgs a = gemm_args(dy16, dt, L.qkv, Bw, dq16, dt, EPI_DEC_QKV); a.scale = KQscale; a.has_scale = 1; a.n_ctx = dt;
          a.C2 = sk; a.C3 = sv; a.ldc2 = kv_ld; a.pos_ptr = kvpos; a.pos_stride = (int)(sizeof(SkwSeqState) / sizeof(int));
          gemm_ln(a, L.qkv, L.attn_ln, l > 0 ? 3 * (l - 1) + 2 : -1, l == 0 && embed_ln); }
        { ProfScope p_(c, PC_DEC_ATTN, 0, 4.0 * live * (pos + 1) * dt);
        skw_dec_self_attn(dq16, sk, sv, &st[0].cur_pos, Bw, H, dt, ntc, datt16, &st[0].active, s, 0, SkwQ8Out{nullptr, nullptr, nullptr, 0}, seqp, c->precision == SKW_PRECISION_F16_MFMA, sa_frag);
        }
        { SkwGemmArgs a = gemm_args(datt16, dt, L.o, Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; a.a_frag = sa_frag; gemm_s(a); }
        {
            { SkwGemmArgs a = gemm_args(dy16, dt, L.cq, Bw, dq16, dt, EPI_F16_PLAIN); a.scale = KQscale; a.has_scale = 1; gemm_ln(a, L.cq, L.cross_ln, 3 * l, false); }
            const bool xp_mq = skw_sw(SW_PROMPT_XATTN_MQ) != 0;
            if (prefill && xp_mq && c->precision == SKW_PRECISION_F16_MFMA && c->pf_nseq > 0) {
                // the prompt pass in the tolerance precision: one read of a sequence's cross K / V^T for up to 128 of its prompt tokens (the encoder attention kernel with the
                // prompt tokens as queries) instead of one per token — 4.6 MB per row per layer otherwise.  The exact precision keeps the single-query kernel: bit-identical to stepping.
                ProfScope p_(c, PC_DEC_XATTN, 4.0 * Bw * (double)nc * 64.0 * H, 4.0 * c->pf_nseq * (double)nc * dt);
                skw_xattn_prefill16(dq16, ck, cv, datt16, c->pf_nseq, c->pf_nq_max, c->pf_meta, c->pf_meta + c->max_batch, c->pf_meta + 2 * c->max_batch, H, dt, nc, c->Tpad, s, c->kv_frag(), xa_frag);
            } else
            { ProfScope p_(c, PC_DEC_XATTN, 4.0 * live * (double)nc * dt, 4.0 * live * (double)nc * dt, true);
            skw_dec_cross_attn_vt(dq16, ck, cv, Bw, H, dt, nc, c->Tpad, datt16, &st[0].active, s, 0, c->kv_frag() ? 2 : c->precision == SKW_PRECISION_F16_MFMA, seqp, p_.ev_a(), p_.ev_b(), xa_frag,
                                  prefill ? nullptr : kclk_node(c, c->cur_group, l));
            }
        }
        { SkwGemmArgs a = gemm_args(datt16, dt, L.co, Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; a.a_frag = xa_frag; gemm_s(a); }
        // f16_mfma, small-M kernels on both sides: fc1 leaves its output as the fragment-order A image fc2 reads (fc2 7.4 -> 6.5 us per launch); the prompt pass's big-tile GEMMs keep rows
        const bool h_frag = afrag_on && c->precision == SKW_PRECISION_F16_MFMA && !bigM && (dt & 127) == 0;
        { SkwGemmArgs a = gemm_args(dy16, dt, L.fc1, Bw, dh16, 4L * dt, EPI_GELU_F16_KPERM); a.gelu_tab = m->gelu_tab; a.c_frag = h_frag; gemm_ln(a, L.fc1, L.mlp_ln, 3 * l + 1, false); }
        { SkwGemmArgs a = gemm_args(dh16, 4L * dt, L.fc2, Bw, dx, dt, EPI_F32); a.res = dx; a.ldres = dt; a.a_frag = h_frag; gemm_s(a); }
    }
    if (want_logits) {
        // (the vocabulary kernel's 256 workgroups would each normalise all 64 rows: measured +6.4 us against this 5.0 us launch)
        { ProfScope p_(c, PC_LAYERNORM, 0, 6.0 * Bw * dt); skw_layernorm(dx, Bw, dt, m->d_ln.w, m->d_ln.b, dy16, nullptr, s); }
        SkwGemmArgs a = gemm_args(dy16, dt, m->te, Bw, c->logits + (size_t)r0 * hp.n_vocab, hp.n_vocab, EPI_F32); GEMM_S(c, a, a.K);
    }
    c->cur = c->stream;
}

// One generation step of a row group as an executable graph: decoder step (positions and tokens read from the device state),
// logit filters + sampling, and the read-back of the group's active count.  Captured once per (group, rows, filter params).
static hipGraphExec_t step_graph(skw_ctx* c, int g, int r0, int n, const SkwLogitParams& lp) {
    for (size_t i = 0; i < c->step_graphs.size(); ++i) {
        auto& sg = c->step_graphs[i];
        if (sg.g == g && sg.r0 == r0 && sg.n == n && sg.precision == c->precision && sg.ln_stats == c->ln_stats_on && sg.kclk == c->kclk_on && sg.sw_epoch == skw_sw_epoch() &&
            memcmp(&sg.lp, &lp, sizeof lp) == 0) {
            if (i + 1 != c->step_graphs.size()) { auto hit = sg; c->step_graphs.erase(c->step_graphs.begin() + i); c->step_graphs.push_back(hit); }   // most recently used last
            return c->step_graphs.back().exec;
        }
    }
    const int NV = c->m->hp.n_vocab; hipStream_t s = c->gstream[g]; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
    c->cur_group = g;
    run_decoder_step(c, r0, n, 0, true, s);
    c->cur_group = 0;
    skw_dec_sample(c->logits + (size_t)r0 * NV, c->static_mask, lp, c->st + r0, c->toks + (size_t)r0 * c->max_tok, c->max_tok, n, c->d_row_live + r0,
        c->probs + (size_t)r0 * skw_probs_row_floats(NV), c->rng, c->clip_idx + r0, c->prompt_buf + (size_t)r0 * SKW_PROMPT_CAP, s);
    if (hipStreamEndCapture(s, &graph) != hipSuccess || !graph) return nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
    hipGraphDestroy(graph);
    if (exec) {
        // bounded: a long-lived server with ragged batches would otherwise keep one executable graph per (group, rows, params) forever
        if (c->step_graphs.size() >= 24) { hipGraphExecDestroy(c->step_graphs.front().exec); c->step_graphs.erase(c->step_graphs.begin()); }
        skw_ctx::StepGraph sg; sg.g = g; sg.r0 = r0; sg.n = n; sg.precision = c->precision; sg.ln_stats = c->ln_stats_on; sg.kclk = c->kclk_on;
         sg.sw_epoch = skw_sw_epoch(); sg.lp = lp; sg.exec = exec; c->step_graphs.push_back(sg);
    }
    return exec;
}

static void build_static_mask(skw_ctx* c, const skw_full_params* p) {
    if (c->static_mask_nst == (p->suppress_nst ? 1 : 0)) return;
    skw_model* m = c->m; std::vector<uint8_t> mask(m->hp.n_vocab, 0);
    mask[m->tok_not] = 1; mask[m->tok_sot] = 1; mask[m->tok_nosp] = 1; mask[m->tok_solm] = 1; mask[m->tok_translate] = 1; mask[m->tok_transcribe] = 1; mask[m->tok_prev] = 1;
    for (int i = 0; i < m->n_lang; ++i) mask[m->tok_sot + 1 + i] = 1;
    if (p->suppress_nst) { for (int id : m->nst_ids) mask[id] = 1; if (m->tok_sp_dash >= 0) mask[m->tok_sp_dash] = 1; if (m->tok_sp_quote >= 0) mask[m->tok_sp_quote] = 1; }
    std::vector<uint8_t> packed(skw_static_mask_bytes(m->hp.n_vocab)); skw_static_mask_pack(mask.data(), m->hp.n_vocab, packed.data());
    hipMemcpyAsync(c->static_mask, packed.data(), packed.size(), hipMemcpyHostToDevice, c->stream); hipStreamSynchronize(c->stream);
    c->static_mask_nst = p->suppress_nst ? 1 : 0;
}

// the sampler's view of the model's special tokens and of the whisper_full_params that shape whisper_process_logits
static SkwLogitParams make_logit_params(const skw_model* m, const skw_full_params* p) {
    const skw_hparams& hp = m->hp;
    SkwLogitParams lp{};
    lp.n_vocab = hp.n_vocab;
    lp.tok_eot = m->tok_eot; lp.tok_sot = m->tok_sot; lp.tok_translate = m->tok_translate; lp.tok_transcribe = m->tok_transcribe; lp.tok_solm = m->tok_solm;
    lp.tok_prev = m->tok_prev; lp.tok_nosp = m->tok_nosp; lp.tok_not = m->tok_not; lp.tok_beg = m->tok_beg;
    lp.n_lang = m->n_lang; lp.tok_space = m->tok_space; lp.tok_sp_dash = m->tok_sp_dash; lp.tok_sp_quote = m->tok_sp_quote;
    lp.suppress_blank = p->suppress_blank; lp.suppress_nst = p->suppress_nst; lp.no_timestamps = p->no_timestamps; lp.single_segment = p->single_segment; lp.max_tokens = p->max_tokens;
    lp.tid0_initial = -1;
    if (p->max_initial_ts > 0.0f) { const float precision = (float)WHISPER_CHUNK_SIZE / hp.n_audio_ctx; lp.tid0_initial = (int)roundf(p->max_initial_ts / precision); }
    lp.n_max = hp.n_text_ctx / 2 - 4;
    return lp;
}

struct SeqAcc { std::vector<skw_segment> seg; std::vector<skw_token> tok; std::string text; };

// whisper_sequence_score: avg_logprobs + entropy of the last 32 tokens
static void sequence_score(const SkwTokenOut* tk, int result_len, double* avg_logprobs, double* entropy) {
    *avg_logprobs = -INFINITY; *entropy = 0.0; if (result_len == 0) return;
    double result = 0.0; for (int i = 0; i < result_len; ++i) result += tk[i].plog;
    *avg_logprobs = result / result_len;
    std::map<int, int> cnts; int cnt = 0; for (int i = std::max(0, result_len - 32); i < result_len; ++i) { cnts[tk[i].id]++; cnt++; }
    double e = 0.0; for (auto& kv : cnts) { double pp = kv.second / (double)cnt; e -= pp * log(pp); } *entropy = e;
}

static int load_clips(skw_ctx* c, const float* const* pcm, const int32_t* n_samples, int n, int on_device, std::vector<int>& n_len, std::vector<int>& n_len_org) {
    char* errbuf = c->errbuf;
    std::vector<long> off(n); std::vector<int> ns(n); n_len.resize(n); n_len_org.resize(n);
    for (int i = 0; i < n; ++i) {
        if (n_samples[i] < 0 || n_samples[i] > c->max_samples) { snprintf(errbuf, 512, "clip %d has %d samples; context was created for at most %d", i, n_samples[i], c->max_samples); return -1; }
        off[i] = (long)i * c->max_samples; ns[i] = n_samples[i];
        n_len[i] = (int)(((long)n_samples[i] + WHISPER_SAMPLE_RATE * 30 + 2 * (WHISPER_N_FFT / 2) - WHISPER_N_FFT) / WHISPER_HOP);
        n_len_org[i] = 1 + (n_samples[i] + WHISPER_N_FFT / 2 - WHISPER_N_FFT) / WHISPER_HOP;
        if (n_samples[i] > 0) HIPCHK(hipMemcpyAsync(c->pcm + off[i], pcm[i], sizeof(float) * n_samples[i], on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->pcm_off, off.data(), sizeof(long) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->n_samples, ns.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->n_len, n_len.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // host vectors go out of scope
    return 0;
}

// cross K/V of window slots src_slot[r] -> dst_slot[r], every layer, 16 bytes per thread (temperature retries keep their encoder pass)
__global__ void k_slot_copy(const half_t* src, long src_layer_stride, half_t* dst, long dst_layer_stride, const int* src_slot, const int* dst_slot, long slot_elems) {
    const int r = blockIdx.y, l = blockIdx.z;
    const uint4* s4 = (const uint4*)(src + (long)l * src_layer_stride + (long)src_slot[r] * slot_elems);
    uint4* d4 = (uint4*)(dst + (long)l * dst_layer_stride + (long)dst_slot[r] * slot_elems);
    const long n16 = slot_elems >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) d4[i] = s4[i];
}
// The windows in `old_slots` (ascending new slot order 0 .. R-1) are decoded again at the next temperature: their cross K/V move to slots
// 0 .. R-1 through a staging copy (a direct move could overwrite another retry's source), instead of running the encoder on them again.
static int move_retry_slots(skw_ctx* c, const std::vector<int>& old_slots) {
    char* errbuf = c->errbuf; const skw_hparams& hp = c->m->hp; const int R = (int)old_slots.size(), L = hp.n_text_layer;
    bool same = true; for (int k = 0; k < R; ++k) same = same && old_slots[k] == k;
    if (same) return 0;
    const long ke = (long)c->kclip(), ka = (long)c->Tpad * hp.n_text_state, ve = (long)hp.n_text_head * 64 * c->Tpad;      // ka: the staging buffer is sized like crossK (Tpad rows per slot)
    if (!c->stageK || !c->stageV || !c->slot_map) {      // all three or none: a partial failure must not leave a later retry launching k_slot_copy on a null buffer
        half_t *sk = nullptr, *sv = nullptr; int* sm = nullptr;
        if (hipMalloc((void**)&sk, (size_t)L * c->max_batch * ka * 2) != hipSuccess || hipMalloc((void**)&sv, (size_t)L * c->max_batch * ve * 2) != hipSuccess
            || hipMalloc((void**)&sm, sizeof(int) * 2 * c->max_batch) != hipSuccess) {
            hipFree(sk); hipFree(sv); hipFree(sm); snprintf(errbuf, 512, "temperature retry: staging buffers for the cross K/V move could not be allocated"); return -1;
        }
        poison_floats(sk, (size_t)L * c->max_batch * ka * 2); poison_floats(sv, (size_t)L * c->max_batch * ve * 2);
        hipFree(c->stageK); hipFree(c->stageV); hipFree(c->slot_map); c->stageK = sk; c->stageV = sv; c->slot_map = sm;
    }
    std::vector<int> h(2 * (size_t)c->max_batch, 0);
    for (int k = 0; k < R; ++k) { h[k] = old_slots[k]; h[c->max_batch + k] = k; }
    HIPCHK(hipMemcpyAsync(c->slot_map, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                                   // (h is a local; retries are rare)
    const int* olds = c->slot_map; const int* news = c->slot_map + c->max_batch;
    const dim3 grid(64, R, L);
    hipLaunchKernelGGL(k_slot_copy, grid, dim3(256), 0, c->stream, c->crossK, (long)c->max_batch * ke, c->stageK, (long)c->max_batch * ke, olds, news, ke);
    hipLaunchKernelGGL(k_slot_copy, grid, dim3(256), 0, c->stream, c->crossV, (long)c->max_batch * ve, c->stageV, (long)c->max_batch * ve, olds, news, ve);
    hipLaunchKernelGGL(k_slot_copy, grid, dim3(256), 0, c->stream, c->stageK, (long)c->max_batch * ke, c->crossK, (long)c->max_batch * ke, news, news, ke);
    hipLaunchKernelGGL(k_slot_copy, grid, dim3(256), 0, c->stream, c->stageV, (long)c->max_batch * ve, c->crossV, (long)c->max_batch * ve, news, news, ve);
    return 0;
}

static int full_batch_impl(skw_ctx* c, const skw_full_params* p, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device, skw_result* results,
                           const int32_t* const* forced_ids, const int32_t* n_forced, std::vector<std::vector<SkwTraceStep>>* traces, uint32_t* const* rng_state = nullptr) {
    char* errbuf = c->errbuf; errbuf[0] = 0;
    WS_READY(c);
    if (n_clips < 1 || n_clips > c->max_batch) { snprintf(errbuf, 512, "n_clips %d outside [1, %d]", n_clips, c->max_batch); return -1; }
    HIPCHK(hipSetDevice(c->m->device));
    const bool tracing = traces != nullptr;
    std::vector<int> f_cursor(n_clips, 0);      // tracing: decisions of clip i made so far (= its position in forced_ids[i])
    if (tracing && !c->trace_dev) {
        int* f = nullptr; SkwTraceStep* t = nullptr;
        if (hipMalloc((void**)&f, sizeof(int) * (size_t)c->max_batch * c->max_tok) != hipSuccess || hipMalloc((void**)&t, sizeof(SkwTraceStep) * (size_t)c->max_batch * c->max_tok) != hipSuccess) {
            hipFree(f); snprintf(errbuf, 512, "trace buffers: device allocation failed"); return -1; }
        c->forced_dev = f; c->trace_dev = t;
    }
    skw_model* m = c->m; const skw_hparams& hp = m->hp; const int NV = hp.n_vocab;
    for (int i = 0; i < n_clips; ++i) { memset(&results[i], 0, sizeof(skw_result)); results[i].min_margin = INFINITY; }
    std::vector<int> n_len, n_len_org;
    if (c->kclk_on && kclk_reset(c)) return -1;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (load_clips(c, pcm, n_samples, n_clips, pcm_on_device, n_len, n_len_org)) return -1;
    build_static_mask(c, p);
    run_mel(c, n_clips);
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    float enc_ms = 0.f, dec_ms = 0.f; int tot_windows = 0, tot_steps = 0, tot_tokens = 0; long tot_row_steps = 0; int used_groups = 1, used_group_rows = 0;

    std::vector<int> seek(n_clips, 0); std::vector<SeqAcc> acc(n_clips);
    // temperature ladder (whisper_full_with_state): per clip, the index of the temperature its current window is decoded at
    std::vector<float> temps; temps.push_back(p->temperature);
    if (p->temperature_inc > 0.0f) for (float t = p->temperature + p->temperature_inc; t < 1.0f + 1e-6f && temps.size() < 16; t += p->temperature_inc) temps.push_back(t);
    std::vector<int> tidx(n_clips, 0), retry_slot(n_clips, -1);   // retry_slot: the window slot whose cross K/V a retrying clip left behind (-1: not retrying)
    // prompt_past (whisper_full_with_state): text already produced in this call conditions the next window of the same clip
    std::vector<std::vector<int>> prompt_past(n_clips); std::vector<int> last_take(n_clips, 0);
    // the sampled passes' generator: whisper.cpp keeps ONE std::mt19937 per state (decoder 0, seeded with 0 when the state is created) and lets it run on across calls.  A caller
    // that owns such a stream per clip (the plugin: one per instance, lib.rs:377-379) hands its state in and gets it back (skw_full_batch_rng); without one the stream starts at
    // seed 0 in every call (D2': what a batch of unrelated clips can do)
    skw_rng_seed(c->rng, n_clips, 0u, c->stream);
    if (rng_state) for (int i = 0; i < n_clips; ++i) if (rng_state[i]) {
        if (rng_state[i][624] > 624u) { snprintf(errbuf, 512, "clip %d: the generator state handed in is not a std::mt19937 state (index %u)", i, rng_state[i][624]); return -1; }
        HIPCHK(hipMemcpyAsync(c->rng + (size_t)i * SKW_RNG_WORDS, rng_state[i], sizeof(uint32_t) * SKW_RNG_WORDS, hipMemcpyHostToDevice, c->stream));
    }
    // language: fixed by the caller, or (lang_id < 0, whisper.cpp's "auto") detected per clip from the [sot] step on the first window
    std::vector<int> lang(n_clips, p->lang_id);
    if (p->lang_id < 0) {
        if (NV < 51865) { snprintf(errbuf, 512, "failed to auto-detect language: the model is not multilingual"); return -3; }
        std::vector<int> act, zero(n_clips, 0); for (int i = 0; i < n_clips; ++i) if (n_len_org[i] > 0) act.push_back(i);
        const int Bd = (int)act.size();
        if (Bd > 0) {
            HIPCHK(hipMemcpyAsync(c->clip_idx, act.data(), sizeof(int) * Bd, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->seek, zero.data(), sizeof(int) * Bd, hipMemcpyHostToDevice, c->stream));
            run_conv(c, Bd); run_encoder(c, Bd, false, true);
            hipLaunchKernelGGL(k_set_tokens, dim3(Bd), dim3(1), 0, c->stream, c->st, m->tok_sot, 0);
            run_decoder_step(c, 0, Bd, 0, true, c->stream);
            hipLaunchKernelGGL(k_lang_argmax, dim3(Bd), dim3(64), 0, c->stream, c->logits, NV, m->tok_sot, m->n_lang, c->row_tok);
            std::vector<int> det(Bd); HIPCHK(hipMemcpyAsync(det.data(), c->row_tok, sizeof(int) * Bd, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
            for (int j = 0; j < Bd; ++j) { lang[act[j]] = det[j]; results[act[j]].lang_id = det[j]; }
        }
        for (int i = 0; i < n_clips; ++i) if (lang[i] < 0) lang[i] = 0;
    } else for (int i = 0; i < n_clips; ++i) results[i].lang_id = p->lang_id;
    int32_t prompt[8]; int n_prompt = 0;
    prompt[n_prompt++] = m->tok_sot;
    if (NV >= 51865) { prompt[n_prompt++] = -1 /* per row: sot + 1 + lang[clip] */; prompt[n_prompt++] = p->translate ? m->tok_translate : m->tok_transcribe; }
    if (p->no_timestamps) prompt[n_prompt++] = m->tok_not;
    SkwLogitParams lp = make_logit_params(m, p);

    while (true) {
        // clips that still have audio to decode ("if only 100ms left, then stop"; "input is too short": delta_min = 10 frames, whisper.cpp #2065)
        // Windows that are decoded again at the next temperature come first: their cross K/V are still in HBM (at last round's slot) and move
        // to slots 0 .. R-1; the encoder then runs only on the new windows, slots R .. Bw-1.
        std::vector<int> act, retry_old;
        for (int i = 0; i < n_clips; ++i) if (retry_slot[i] >= 0) { act.push_back(i); retry_old.push_back(retry_slot[i]); }
        const int R = (int)act.size();
        for (int i = 0; i < n_clips; ++i) if (retry_slot[i] < 0 && n_len_org[i] >= SKW_DELTA_MIN && seek[i] + SKW_DELTA_MIN < n_len_org[i]) act.push_back(i);
        if (act.empty()) break;
        const int Bw = (int)act.size();
        if (R > 0 && move_retry_slots(c, retry_old)) return -1;
        for (int j = 0; j < Bw; ++j) retry_slot[act[j]] = -1;
        std::vector<int> sk(Bw); for (int j = 0; j < Bw; ++j) sk[j] = seek[act[j]];
        HIPCHK(hipMemcpyAsync(c->clip_idx, act.data(), sizeof(int) * Bw, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->seek, sk.data(), sizeof(int) * Bw, hipMemcpyHostToDevice, c->stream));
        // per-row prompt: [prev] + the last n_text_ctx/2 tokens of the clip's prompt_past (passes at t < 0.5 only) + sot, language, task (, notimestamps)
        std::vector<int> pbuf((size_t)Bw * SKW_PROMPT_CAP, 0), np_row(Bw, 0);
        for (int j = 0; j < Bw; ++j) {
            const int ci = act[j]; int* pr = pbuf.data() + (size_t)j * SKW_PROMPT_CAP; int n = 0, take = 0;
            if (!prompt_past[ci].empty() && temps[tidx[ci]] < 0.5f) {
                // the last bound only binds with no_timestamps: every position stays inside n_text_ctx
                take = std::min(std::min(hp.n_text_ctx / 2, (int)prompt_past[ci].size()), hp.n_text_ctx - lp.n_max - n_prompt - 1);
                pr[n++] = m->tok_prev; for (int i = 0; i < take; ++i) pr[n++] = prompt_past[ci][prompt_past[ci].size() - take + i];
            }
            last_take[ci] = take;
            for (int t = 0; t < n_prompt; ++t) pr[n++] = prompt[t] >= 0 ? prompt[t] : m->tok_sot + 1 + lang[ci];
            np_row[j] = n;
        }
        HIPCHK(hipMemcpyAsync(c->prompt_buf, pbuf.data(), sizeof(int) * pbuf.size(), hipMemcpyHostToDevice, c->stream));
        lp.any_sampled = 0; for (int j = 0; j < Bw; ++j) if (temps[tidx[act[j]]] > 0.0f) lp.any_sampled = 1;      // (part of the step graph's key: greedy passes run the lean sampler)
        std::vector<int> fbuf;
        if (tracing) {      // this pass's forced tokens per row: the clip's sequence from its cursor on (-1 = none: the row feeds its own choices)
            fbuf.assign((size_t)Bw * c->max_tok, -1);
            if (forced_ids) for (int j = 0; j < Bw; ++j) { const int ci = act[j]; if (!forced_ids[ci]) continue;
                for (int k = 0; k < c->max_tok && f_cursor[ci] + k < n_forced[ci]; ++k) fbuf[(size_t)j * c->max_tok + k] = forced_ids[ci][f_cursor[ci] + k]; }
            HIPCHK(hipMemcpyAsync(c->forced_dev, fbuf.data(), sizeof(int) * fbuf.size(), hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(hipEventRecord(c->ev[2], c->stream));
        if (R < Bw) { run_conv(c, Bw, R); run_encoder(c, Bw, false, true, R); }
        HIPCHK(hipEventRecord(c->ev[3], c->stream));
        // decoder state
        for (int j = 0; j < Bw; ++j) {
            SkwSeqState& s = c->h_st[j]; memset(&s, 0, sizeof s);
            s.active = 1; s.seek_delta = 100 * WHISPER_CHUNK_SIZE; s.seek = sk[j];
            s.seek_end = n_len_org[act[j]]; s.n_prompt = np_row[j]; s.min_margin = INFINITY; s.cur_token = pbuf[(size_t)j * SKW_PROMPT_CAP]; s.cur_pos = 0;
            s.temperature = temps[tidx[act[j]]];
        }
        // The prompt in one pass (whisper.cpp evaluates it in one whisper_decode call): every prompt token but a row's last becomes a row of ONE decoder pass —
        // the step's own kernels, a row per (sequence, position), the caches addressed through the row's sequence — which fills the self-attention K / V
        // of those positions in every layer; the rows then start at their last prompt token, whose logits the first sampling step needs.  A window of a long
        // file carries up to 224 tokens of previous text in its prompt: 225 steps of 1.4 ms become one pass.  Per-row arithmetic does not depend on what else is
        // in the batch, so the results are the stepped ones, bit for bit (tests/test_gpu_parity.py).
        int prefill_passes = 0;
        if (c->prompt_pass_on) {
            const int ntc = hp.n_text_ctx;
            std::vector<SkwSeqState> pf; std::vector<int> row_end;      // row_end: cumulative rows after each sequence (chunks are whole sequences)
            for (int j = 0; j < Bw; ++j) {
                for (int k = 0; k + 1 < np_row[j]; ++k) { SkwSeqState t; memset(&t, 0, sizeof t);
                t.active = 1; t.cur_token = pbuf[(size_t)j * SKW_PROMPT_CAP + k]; t.cur_pos = k; t.pad = j; t.seek = j * ntc + k; pf.push_back(t); }
                row_end.push_back((int)pf.size());
                c->h_st[j].cur_token = pbuf[(size_t)j * SKW_PROMPT_CAP + np_row[j] - 1]; c->h_st[j].cur_pos = np_row[j] - 1;
            }
            size_t lo = 0; int jlo = 0;
            while (lo < pf.size()) {
                int jhi = jlo; while (jhi < Bw && row_end[jhi] - (int)lo <= c->rows_cap) ++jhi;      // sequences jlo .. jhi-1 fit (a prompt is at most SKW_PROMPT_CAP - 1 <= rows_cap rows)
                const size_t hi = row_end[jhi - 1]; const int n = (int)(hi - lo);
                if (n > 0) {
                    HIPCHK(hipMemcpyAsync(c->pf_st, pf.data() + lo, sizeof(SkwSeqState) * n, hipMemcpyHostToDevice, c->stream));
                    std::vector<int> meta((size_t)3 * c->max_batch, 0); int ns = 0, nqmax = 0;
                    for (int j = jlo; j < jhi; ++j) { const int r0j = (j ? row_end[j - 1] : 0) - (int)lo, nqj = row_end[j] - (j ? row_end[j - 1] : 0); if (nqj <= 0) continue;
                        meta[ns] = r0j; meta[c->max_batch + ns] = nqj; meta[2 * c->max_batch + ns] = j; nqmax = std::max(nqmax, nqj); ++ns; }
                    HIPCHK(hipMemcpyAsync(c->pf_meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, c->stream));
                    c->pf_nseq = ns; c->pf_nq_max = nqmax;
                    run_decoder_step(c, 0, n, 0, false, c->stream, true);
                    c->pf_nseq = 0;
                    HIPCHK(hipStreamSynchronize(c->stream));      // (pf is a host vector and pf_st is reused by the next chunk)
                    ++prefill_passes;
                }
                lo = hi; jlo = jhi;
            }
        }
        HIPCHK(hipMemcpyAsync(c->st, c->h_st, sizeof(SkwSeqState) * Bw, hipMemcpyHostToDevice, c->stream));
        // row groups: G contiguous ranges of the window batch, each on its own stream (one group while profiling, so kernel times do not overlap)
        const bool profiling = c->prof && c->prof->on;
        // Default: ONE row group in both precisions.  Rounds 3-4 shipped two groups for f16_mfma at >= 64 rows (+1.3 % on the step in a same-box A/B, profiles/r04g).  Round 5's in-kernel
        // launch clock (skw_ctx_kernel_clock) showed what that bought and cost: the groups' cross-attention launches overlap for 27 % of their time and stretch each other from 28 to
        // 35-37 us, a profiler serialises the two streams (its per-kernel averages then describe launches that never ran: decode 129 -> 197 ms under rocprofv3), and the dispatch
        // front end is a serial resource — two chains of small kernels on two streams gain ~20 % over one (tools/probe/probe_stream_overlap.hip) while the launch count doubles.
        // Same-box, round 5 (profiles/r05d): 167.13 ms (two groups) vs 167.59 ms (one) per 64-clip step — inside the +-1.5 % two builds differ by — with the dominant kernel at
        // 0.47 of HBM peak per launch (0.66 while any launch is in flight) against 0.77 as one group.  One group: clock, HIP events and rocprofv3 agree on every launch within 2 %.
        // The exact precision loses 3 % with two groups, smaller batches more (16-row launches).  SKW_DECODE_GROUPS=n overrides.
        const int n_groups = c->n_groups > 0 ? c->n_groups : 1;
        // (f16_mfma, one group: eager launches 8 steps ahead measure the same 179 ms; the graph leaves the host idle, which matters with eight ranks on one node)
        const bool use_graphs = c->use_graphs >= 0 ? c->use_graphs != 0 : true;
        // Groups are cut at multiples of 16 rows: the f16_mfma step keeps its attention / FC1 outputs as fragment-order images (skw_afrag_off), which
        // scatter a group's rows over whole 16-row tiles of its scratch — two groups sharing a tile would overwrite each other.
        int g_r0[skw_ctx::MAX_GROUPS], g_n[skw_ctx::MAX_GROUPS]; bool g_live[skw_ctx::MAX_GROUPS];
        int G = 0;
        {
            const int Gw = std::max(1, std::min(n_groups, Bw / 8));
            int prev = 0;
            for (int g = 1; g <= Gw; ++g) {
                const int cut = g == Gw ? Bw : std::min(Bw, (int)(((long)Bw * g / Gw + 8) & ~15L));
                if (cut > prev) { g_r0[G] = prev; g_n[G] = cut - prev; g_live[G] = true; ++G; prev = cut; }
            }
        }
        used_groups = G; used_group_rows = g_n[0];
        for (int j = 0; j < Bw; ++j) c->h_row_live[j] = 1;      // (every kernel of the previous window has drained: c->stream was synchronised at its end)
        HIPCHK(hipEventRecord(c->ev[5], c->stream));
        for (int g = 0; g < G; ++g) HIPCHK(hipStreamWaitEvent(c->gstream[g], c->ev[5], 0));
        // every step = decoder step (token and position from the device state) + k_dec_sample, which feeds the next prompt token
        // while a row is still inside its prompt and samples afterwards; rows have prompts of different lengths
        hipGraphExec_t gexec[skw_ctx::MAX_GROUPS] = {};
        if (use_graphs && !profiling && !tracing) for (int g = 0; g < G; ++g) gexec[g] = step_graph(c, g, g_r0[g], g_n[g], lp);   // nullptr -> eager launches
        auto sample = [&](int g) {
            c->cur = c->gstream[g];
            { ProfScope p_(c, PC_DEC_SAMPLE, 0, 4.0 * g_n[g] * NV); skw_dec_sample(c->logits + (size_t)g_r0[g] * NV, c->static_mask, lp, c->st + g_r0[g],
                c->toks + (size_t)g_r0[g] * c->max_tok, c->max_tok, g_n[g], c->d_row_live + g_r0[g], c->probs + (size_t)g_r0[g] * skw_probs_row_floats(NV), c->rng,
                c->clip_idx + g_r0[g], c->prompt_buf + (size_t)g_r0[g] * SKW_PROMPT_CAP, c->gstream[g],
                                                                                     tracing ? c->forced_dev + (size_t)g_r0[g] * c->max_tok : nullptr,
                                                                                         tracing ? c->trace_dev + (size_t)g_r0[g] * c->max_tok : nullptr); }
            c->cur = c->stream;
            return hipSuccess;
        };
        // The host learns whether a group still has live rows only by waiting for its stream, and a wait + relaunch per step leaves the
        // GPU idle for the turnaround.  So steps are enqueued `ahead` at a time and the count is read once per block: a step that
        // runs after the last row of its group has finished changes nothing (its attention and sampling kernels return at once for
        // finished rows; what the GEMMs write for them is scratch), it only costs its launches.
        const int step_cap = SKW_PROMPT_CAP + lp.n_max + 2;
        const int ahead = profiling ? 1 : 8;
        for (int i = 0; i < step_cap; i += ahead) {
            const int nstep = std::min(ahead, step_cap - i);
            for (int k = 0; k < nstep; ++k)
                for (int g = 0; g < G; ++g) if (g_live[g]) {
                    if (gexec[g]) HIPCHK(hipGraphLaunch(gexec[g], c->gstream[g]));
                    else {
                        // (ahead == 1: the previous step has drained)
                        if (profiling) { int lv = 0; for (int j = g_r0[g]; j < g_r0[g] + g_n[g]; ++j) lv += ((volatile int*)c->h_row_live)[j] != 0; c->live_rows_hint = lv; }
                        c->cur_group = g;
                        run_decoder_step(c, g_r0[g], g_n[g], i + k, true, c->gstream[g]); c->live_rows_hint = -1; c->cur_group = 0; HIPCHK(sample(g));
                    }
                }
            bool any = false;
            for (int g = 0; g < G; ++g) if (g_live[g]) { HIPCHK(hipStreamSynchronize(c->gstream[g]));
            int live = 0; for (int j = g_r0[g]; j < g_r0[g] + g_n[g]; ++j) live += ((volatile int*)c->h_row_live)[j] != 0;
                                                        if (live <= 0) g_live[g] = false; else any = true; }
            if (!any) break;
        }
        for (int g = 0; g < G; ++g) { HIPCHK(hipEventRecord(c->gev[g], c->gstream[g])); HIPCHK(hipStreamWaitEvent(c->stream, c->gev[g], 0)); }
        HIPCHK(hipMemcpyAsync(c->h_st, c->st, sizeof(SkwSeqState) * Bw, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_toks, c->toks, sizeof(SkwTokenOut) * Bw * c->max_tok, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->ev[4], c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (tracing) {
            std::vector<SkwTraceStep> tb((size_t)Bw * c->max_tok);
            HIPCHK(hipMemcpy(tb.data(), c->trace_dev, sizeof(SkwTraceStep) * tb.size(), hipMemcpyDeviceToHost));
            for (int j = 0; j < Bw; ++j) {
                const int ci = act[j], n = std::min(c->h_st[j].n_tokens, c->max_tok);
                if (forced_ids && forced_ids[ci] && f_cursor[ci] + n > n_forced[ci]) { snprintf(errbuf, 512,
                    "clip %d: forced token sequence exhausted (%d given, decision %d reached): the runs' control flow diverged", ci, n_forced[ci], f_cursor[ci] + n);
                return -4; }
                (*traces)[ci].insert((*traces)[ci].end(), tb.begin() + (size_t)j * c->max_tok, tb.begin() + (size_t)j * c->max_tok + n);
                f_cursor[ci] += n;
            }
        }
        { float a = 0, b = 0; hipEventElapsedTime(&a, c->ev[2], c->ev[3]); hipEventElapsedTime(&b, c->ev[3], c->ev[4]); enc_ms += a; dec_ms += b; }
        tot_windows += Bw;
        // decoder passes until the last row finished (the prompt pass counts as one); row-steps: (row, token) pairs that streamed cross K / V, prompt tokens included
        { int mx = 0; for (int j = 0; j < Bw; ++j) { const int rs = c->h_st[j].n_prompt - 1 + c->h_st[j].n_tokens;
        mx = std::max(mx, c->prompt_pass_on ? c->h_st[j].n_tokens : rs); tot_row_steps += rs; } tot_steps += mx + prefill_passes; }
        // per-clip: ranking, segment assembly, seek update (whisper_full_with_state tail)
        for (int j = 0; j < Bw; ++j) {
            const int ci = act[j]; const SkwSeqState& s = c->h_st[j]; const SkwTokenOut* tk = c->h_toks + (size_t)j * c->max_tok; skw_result& R = results[ci]; SeqAcc& A = acc[ci];
            if (tidx[ci] == 0) R.n_windows++;
            R.n_decode_steps += 1 + std::max(0, s.n_tokens - 1);     // the prompt is one decode call in whisper.cpp, then one per sampled token but the last
            bool failed = s.failed != 0; int n_tok = s.n_tokens; const int result_len = s.result_len;
            double avg_logprobs = -INFINITY, entropy = 0.0;
            if (!failed) { n_tok = result_len; sequence_score(tk, result_len, &avg_logprobs, &entropy); if (result_len > 32 && entropy < p->entropy_thold) failed = true; }
            if (s.min_margin < R.min_margin) R.min_margin = s.min_margin;
            if (failed || (avg_logprobs < p->logprob_thold && s.no_speech_prob < p->no_speech_thold)) {
                R.fallback_requested++;
                if (tidx[ci] + 1 < (int)temps.size()) { tidx[ci]++; retry_slot[ci] = j; continue; }   // this window again, at the next temperature, on the cross K/V it already has (slot j)
            }
            tidx[ci] = 0;
            int seek_delta = s.seek_delta;
            const bool is_no_speech = (s.no_speech_prob > p->no_speech_thold && avg_logprobs < p->logprob_thold);
            {   // update prompt_past: what this window's prompt took from it is always kept; this window's tokens only when it is speech
                std::vector<int>& pp = prompt_past[ci]; std::vector<int> keep(pp.end() - last_take[ci], pp.end());
                pp = keep;
                if (!is_no_speech) for (int i = 0; i < result_len; ++i) pp.push_back(tk[i].id);
            }
            if (n_tok > 0 && !is_no_speech) {
                int i0 = 0; int64_t t0 = seek[ci] + 2 * (tk[0].tid - m->tok_beg); std::string text;
                auto push = [&](int64_t a, int64_t b, int from, int to) {
                    skw_segment sg{}; sg.t0 = a; sg.t1 = b; sg.tok_begin = (int)A.tok.size();
                    for (int q = from; q < to; ++q) { skw_token o{tk[q].id, tk[q].tid, tk[q].p, tk[q].plog, tk[q].pt, tk[q].ptsum, tk[q].margin}; A.tok.push_back(o); }
                    sg.tok_end = (int)A.tok.size(); sg.text_off = (int)A.text.size(); sg.text_len = (int)text.size(); A.text += text; A.seg.push_back(sg);
                };
                for (int i = 0; i < n_tok; ++i) {
                    if (tk[i].id < m->tok_eot) text += m->tok_str[tk[i].id];
                    if (tk[i].id > m->tok_beg && !p->single_segment) {
                        const int64_t t1 = seek[ci] + 2 * (tk[i].tid - m->tok_beg);
                        if (!text.empty()) push(t0, t1, i0, i + 1);
                        text.clear();
                        while (i < n_tok && tk[i].id > m->tok_beg) i++;
                        i--; t0 = t1; i0 = i + 1;
                    }
                }
                if (!text.empty()) push(t0, seek[ci] + seek_delta, i0, n_tok);
            }
            const bool single_timestamp_ending = n_tok > 1 && tk[n_tok - 2].id < m->tok_beg && tk[n_tok - 1].id > m->tok_beg;
            if (single_timestamp_ending) seek_delta = std::min(n_len_org[ci] - seek[ci], WHISPER_CHUNK_SIZE * 100);
            seek[ci] += seek_delta;
        }
    }
    HIPCHK(hipEventRecord(c->ev[5], c->stream));
    if (rng_state) for (int i = 0; i < n_clips; ++i) if (rng_state[i])
        HIPCHK(hipMemcpyAsync(rng_state[i], c->rng + (size_t)i * SKW_RNG_WORDS, sizeof(uint32_t) * SKW_RNG_WORDS, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());          // a launch that failed (bad configuration, lost device) must not look like a transcript
    for (int i = 0; i < n_clips; ++i) {
        skw_result& R = results[i]; SeqAcc& A = acc[i];
        R.n_segments = (int)A.seg.size(); R.n_tokens = (int)A.tok.size(); R.text_len = (int)A.text.size();
        R.segments = (skw_segment*)malloc(sizeof(skw_segment) * std::max<size_t>(1, A.seg.size())); memcpy(R.segments, A.seg.data(), sizeof(skw_segment) * A.seg.size());
        R.tokens = (skw_token*)malloc(sizeof(skw_token) * std::max<size_t>(1, A.tok.size())); memcpy(R.tokens, A.tok.data(), sizeof(skw_token) * A.tok.size());
        R.text = (char*)malloc(A.text.size() + 1); memcpy(R.text, A.text.data(), A.text.size()); R.text[A.text.size()] = 0;
        tot_tokens += R.n_tokens;
    }
    prof_collect(c);
    { float a = 0, t = 0; hipEventElapsedTime(&a, c->ev[0], c->ev[1]); hipEventElapsedTime(&t, c->ev[0], c->ev[5]);
      c->timing.mel_ms = a; c->timing.encode_ms = enc_ms; c->timing.decode_ms = dec_ms;
      c->timing.total_ms = t; c->timing.n_windows = tot_windows; c->timing.n_decode_steps = tot_steps;
      c->timing.n_tokens = tot_tokens; c->timing.n_row_steps = (int32_t)tot_row_steps; c->timing.decode_groups = used_groups; c->timing.decode_group_rows = used_group_rows; }
    return 0;
}
extern "C" int skw_full_batch(skw_ctx* c, const skw_full_params* p, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device, skw_result* results) {
    try { return full_batch_impl(c, p, pcm, n_samples, n_clips, pcm_on_device, results, nullptr, nullptr, nullptr); }
    catch (const std::exception& e) { snprintf(c->errbuf, 512, "skw_full_batch: %s", e.what()); return -5; }      // nothing may unwind across the C ABI
}
extern "C" void skw_rng_state_init(uint32_t* state) {      // std::mt19937(0), as whisper_init_state leaves decoder 0's generator
    state[0] = 0u; for (int i = 1; i < 624; ++i) state[i] = 1812433253u * (state[i - 1] ^ (state[i - 1] >> 30)) + (uint32_t)i;
    state[624] = 624u;
}
extern "C" int skw_full_batch_rng(skw_ctx* c, const skw_full_params* p, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device,
                                  uint32_t* const* rng_state, skw_result* results) {
    try { return full_batch_impl(c, p, pcm, n_samples, n_clips, pcm_on_device, results, nullptr, nullptr, nullptr, rng_state); }
    catch (const std::exception& e) { snprintf(c->errbuf, 512, "skw_full_batch_rng: %s", e.what()); return -5; }
}
extern "C" int skw_full_batch_traced(skw_ctx* c, const skw_full_params* p, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device,
                                     const int32_t* const* forced_ids, const int32_t* n_forced, skw_trace* traces, skw_result* results) {
    static_assert(sizeof(skw_trace_step) == sizeof(SkwTraceStep), "skw_trace_step is SkwTraceStep");
    try {
        if (!traces || (forced_ids && !n_forced)) { snprintf(c->errbuf, 512, "skw_full_batch_traced: bad arguments"); return -1; }
        for (int i = 0; i < n_clips; ++i) { traces[i].n = 0; traces[i].steps = nullptr; }
        std::vector<std::vector<SkwTraceStep>> tr(std::max(0, n_clips));
        const int rc = full_batch_impl(c, p, pcm, n_samples, n_clips, pcm_on_device, results, forced_ids, n_forced, &tr);
        if (rc) return rc;
        for (int i = 0; i < n_clips; ++i) {
            traces[i].n = (int32_t)tr[i].size(); traces[i].steps = (skw_trace_step*)malloc(sizeof(skw_trace_step) * std::max<size_t>(1, tr[i].size()));
            memcpy(traces[i].steps, tr[i].data(), sizeof(skw_trace_step) * tr[i].size());
        }
        return 0;
    } catch (const std::exception& e) { snprintf(c->errbuf, 512, "skw_full_batch_traced: %s", e.what()); return -5; }
}
// Test hook (tests/test_gpu_logit_rules.py): K11 on its own.  n_rows decoders whose tokens sampled so far in their window are hist[r][0 .. n_hist[r]) meet
// caller-supplied logits; ONE launch of the sampler (form 0: the one the decode step uses; form 1: the streaming kernel, which leaves the filtered row in
// memory) makes each row's next decision.  has_ts / seek_delta / result_len are replayed from the history by the token loop's update rule, as
// oracle/skwo_debug_process_logits does.  Outputs per row: the decision (skw_token), its trace record, and (form 1, optional) the filtered logits.
extern "C" int skw_debug_sample_rows(skw_ctx* c, const skw_full_params* p, int n_rows, const int32_t* hist, int hist_stride, const int32_t* n_hist, const float* logits_host, int form,
                                     float* filtered_out, skw_token* tok_out, skw_trace_step* trace_out) {
    char* errbuf = c->errbuf; errbuf[0] = 0;
    WS_READY(c);
    if (n_rows < 1 || n_rows > c->max_batch) { snprintf(errbuf, 512, "n_rows %d outside [1, %d]", n_rows, c->max_batch); return -1; }
    HIPCHK(hipSetDevice(c->m->device));
    const skw_model* m = c->m; const int NV = m->hp.n_vocab, MT = c->max_tok;
    for (int r = 0; r < n_rows; ++r) if (n_hist[r] < 0 || n_hist[r] >= MT || n_hist[r] > hist_stride) { snprintf(errbuf, 512, "row %d: history of %d tokens (at most %d)", r, n_hist[r], MT - 1);
    return -1; }
    if (!c->trace_dev) {
        int* f = nullptr; SkwTraceStep* t = nullptr;
        if (hipMalloc((void**)&f, sizeof(int) * (size_t)c->max_batch * MT) != hipSuccess || hipMalloc((void**)&t, sizeof(SkwTraceStep) * (size_t)c->max_batch * MT) != hipSuccess) {
            hipFree(f); snprintf(errbuf, 512, "trace buffers: device allocation failed"); return -1; }
        c->forced_dev = f; c->trace_dev = t;
    }
    build_static_mask(c, p);
    SkwLogitParams lp = make_logit_params(m, p); lp.any_sampled = 0;
    std::vector<SkwTokenOut> toks((size_t)n_rows * MT); memset(toks.data(), 0, toks.size() * sizeof(SkwTokenOut));
    std::vector<int> forced((size_t)n_rows * MT, -1), zero(n_rows, 0);
    for (int r = 0; r < n_rows; ++r) {
        SkwSeqState& s = c->h_st[r]; memset(&s, 0, sizeof s);
        s.active = 1; s.seek_delta = 100 * WHISPER_CHUNK_SIZE; s.seek = 0; s.seek_end = 100 * WHISPER_CHUNK_SIZE; s.n_prompt = 1; s.min_margin = INFINITY;
        s.cur_pos = n_hist[r]; s.n_tokens = n_hist[r];
        for (int i = 0; i < n_hist[r]; ++i) {
            const int id = hist[(size_t)r * hist_stride + i]; toks[(size_t)r * MT + i].id = id;
            if (id > m->tok_beg) {
                const int sd = 2 * (id - m->tok_beg);
                if (s.has_ts && s.seek_delta > sd && s.result_len < i) { snprintf(errbuf, 512, "row %d: the token loop cannot produce this history (timestamp goes backwards at %d)", r, i);
                return -1; }
                s.seek_delta = sd; s.result_len = i + 1; s.has_ts = 1;
            }
        }
        c->h_row_live[r] = 1;
    }
    HIPCHK(hipMemcpyAsync(c->st, c->h_st, sizeof(SkwSeqState) * n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->toks, toks.data(), sizeof(SkwTokenOut) * toks.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->forced_dev, forced.data(), sizeof(int) * forced.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->clip_idx, zero.data(), sizeof(int) * n_rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->logits, logits_host, sizeof(float) * (size_t)n_rows * NV, hipMemcpyHostToDevice, c->stream));
    skw_debug_force_stream_sampler(form == 1);
    skw_dec_sample(c->logits, c->static_mask, lp, c->st, c->toks, MT, n_rows, c->d_row_live, c->probs, c->rng, c->clip_idx, c->prompt_buf, c->stream, c->forced_dev, c->trace_dev);
    skw_debug_force_stream_sampler(0);
    HIPCHK(hipGetLastError());
    std::vector<SkwTraceStep> tr((size_t)n_rows * MT);
    HIPCHK(hipMemcpyAsync(toks.data(), c->toks, sizeof(SkwTokenOut) * toks.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(tr.data(), c->trace_dev, sizeof(SkwTraceStep) * tr.size(), hipMemcpyDeviceToHost, c->stream));
    if (filtered_out) HIPCHK(hipMemcpyAsync(filtered_out, c->logits, sizeof(float) * (size_t)n_rows * NV, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int r = 0; r < n_rows; ++r) {
        const SkwTokenOut& t = toks[(size_t)r * MT + n_hist[r]];
        if (tok_out) { skw_token o; o.id = t.id; o.tid = t.tid; o.p = t.p; o.plog = t.plog; o.pt = t.pt; o.ptsum = t.ptsum; o.margin = t.margin; tok_out[r] = o; }
        if (trace_out) memcpy(&trace_out[r], &tr[(size_t)r * MT + n_hist[r]], sizeof(skw_trace_step));
    }
    return 0;
}
extern "C" void skw_trace_free(skw_trace* t) { if (!t) return; free(t->steps); t->steps = nullptr; t->n = 0; }
extern "C" void skw_result_free(skw_result* r) { if (!r) return; free(r->segments); free(r->tokens); free(r->text); memset(r, 0, sizeof *r); }

// ------------------------------------------------------------------ stage taps
__global__ void k_transpose_mel(const float* mel, int n_len, int n_mel, float* out) {   // [frame][mel] -> [mel][frame]
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long)n_len * n_mel) return; int fr = (int)(i / n_mel), j = (int)(i % n_mel); out[(long)j * n_len + fr] = mel[i];
}
extern "C" int skw_log_mel(skw_ctx* c, const float* pcm_host, int n_samples, float* mel_out, size_t cap, int* n_len_o, int* n_len_org_o) {
    char* errbuf = c->errbuf;
    WS_READY(c);
    HIPCHK(hipSetDevice(c->m->device));
    std::vector<int> n_len, n_len_org; const float* pp[1] = {pcm_host}; int32_t ns[1] = {n_samples};
    if (load_clips(c, pp, ns, 1, 0, n_len, n_len_org)) return -1;
    run_mel(c, 1);
    const size_t n = (size_t)n_len[0] * c->m->hp.n_mels;
    if (cap < n) { snprintf(errbuf, 512, "mel_out too small: need %zu floats", n); return -1; }
    float* tmp = nullptr; HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)));
    hipLaunchKernelGGL(k_transpose_mel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->mel, n_len[0], c->m->hp.n_mels, tmp);
    HIPCHK(hipMemcpyAsync(mel_out, tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); hipFree(tmp);
    *n_len_o = n_len[0]; *n_len_org_o = n_len_org[0]; return 0;
}
static int tap_prepare(skw_ctx* c, const float* pcm_host, int n_samples, int seek) {
    char* errbuf = c->errbuf;
    WS_READY(c);
    HIPCHK(hipSetDevice(c->m->device));
    std::vector<int> n_len, n_len_org; const float* pp[1] = {pcm_host}; int32_t ns[1] = {n_samples};
    if (load_clips(c, pp, ns, 1, 0, n_len, n_len_org)) return -1;
    run_mel(c, 1);
    int zero = 0; HIPCHK(hipMemcpyAsync(c->clip_idx, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream)); HIPCHK(hipMemcpyAsync(c->seek, &seek, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    run_conv(c, 1); return 0;
}
extern "C" int skw_conv_stem(skw_ctx* c, const float* pcm_host, int n_samples, int seek, float* x0) {
    char* errbuf = c->errbuf; if (tap_prepare(c, pcm_host, n_samples, seek)) return -1;
    HIPCHK(hipMemcpyAsync(x0, c->x, sizeof(float) * c->m->hp.n_audio_ctx * c->m->hp.n_audio_state, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return 0;
}
// cross V^T [(h*64 + c)][Tpad kperm] of batch slot 0 -> natural [key][d] f32
__global__ void k_vt2f_copy(const half_t* src, float* dst, int nc, int dt, int Tpad) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
if (i >= (long)nc * dt) return; int key = (int)(i / dt), n = (int)(i % dt); dst[i] = (float)src[(long)n * Tpad + skw_kperm(key)]; }
// the same exports from the fragment-order images
__global__ void k_kfrag2f_copy(const half_t* src, float* dst, int nc, int dt, int H, int Tpad) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
if (i >= (long)nc * dt) return; int key = (int)(i / dt), n = (int)(i % dt); dst[i] = (float)src[skw_kfrag_off(0, H, Tpad, key, n & ~7) + (n & 7)]; }
__global__ void k_vtfrag2f_copy(const half_t* src, float* dst, int nc, int dt, int H, int Tpad) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
if (i >= (long)nc * dt) return; int key = (int)(i / dt), n = (int)(i % dt);
const int p = skw_kperm(key); dst[i] = (float)src[skw_vtfrag_off(0, H, Tpad, n, p & ~7) + (p & 7)]; }
__global__ void k_h2f_copy(const half_t* src, float* dst, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = (float)src[i]; }
extern "C" int skw_encode(skw_ctx* c, const float* pcm_host, int n_samples, int seek, float* enc_out, float* cross_k, float* cross_v) {
    char* errbuf = c->errbuf; if (tap_prepare(c, pcm_host, n_samples, seek)) return -1;
    const skw_hparams& hp = c->m->hp; const int nc = hp.n_audio_ctx, d = hp.n_audio_state, dt = hp.n_text_state;
    run_encoder(c, 1, true, true);
    HIPCHK(hipMemcpyAsync(enc_out, c->enc_out32, sizeof(float) * nc * d, hipMemcpyDeviceToHost, c->stream));
    if (cross_k && cross_v) {
        const long n = (long)nc * dt; float* tmp = nullptr; HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)));
        for (int l = 0; l < hp.n_text_layer; ++l) for (int kv = 0; kv < 2; ++kv) {
            const half_t* vsrc = c->crossV + (size_t)l * c->max_batch * hp.n_text_head * 64 * c->Tpad; const half_t* ksrc = c->crossK + (size_t)l * c->max_batch * c->kclip();
            if (kv && c->kv_frag()) hipLaunchKernelGGL(k_vtfrag2f_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, vsrc, tmp, nc, dt, hp.n_text_head, c->Tpad);
            else if (kv) hipLaunchKernelGGL(k_vt2f_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, vsrc, tmp, nc, dt, c->Tpad);
            else if (c->kv_frag()) hipLaunchKernelGGL(k_kfrag2f_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ksrc, tmp, nc, dt, hp.n_text_head, c->Tpad);
            else hipLaunchKernelGGL(k_h2f_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ksrc, tmp, n);
            HIPCHK(hipMemcpyAsync((kv ? cross_v : cross_k) + (size_t)l * n, tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
        }
        hipFree(tmp);
    }
    HIPCHK(hipStreamSynchronize(c->stream)); return 0;
}
extern "C" int skw_decode_logits(skw_ctx* c, const int32_t* tokens, int n_tokens, float* logits) {
    char* errbuf = c->errbuf;
    WS_READY(c);
    HIPCHK(hipSetDevice(c->m->device));
    if (n_tokens < 1 || n_tokens > c->m->hp.n_text_ctx) { snprintf(errbuf, 512, "bad n_tokens"); return -1; }
    for (int t = 0; t < n_tokens; ++t) {
        hipLaunchKernelGGL(k_set_tokens, dim3(1), dim3(1), 0, c->stream, c->st, tokens[t], t);
        run_decoder_step(c, 0, 1, t, t == n_tokens - 1, c->stream);
    }
    HIPCHK(hipMemcpyAsync(logits, c->logits, sizeof(float) * c->m->hp.n_vocab, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return 0;
}

// ------------------------------------------------------------------ arithmetic-contract probes (tests/test_gpu_math.py)
__global__ void k_math_probe(int kind, const float* in, float* out, long n, const uint16_t* gelu_tab) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    float x = in[i], y;
    if (kind == 0) y = skw_expf(x);
    else if (kind == 1) y = skw_logf(x);
    else if (kind == 2) y = (float)((half_t)x);
    else if (kind == 3) y = skw_round_f16(x);
    else if (kind == 4) { if (x <= -10.0f) y = 0.0f; else if (x >= 10.0f) y = x; else { half_t h = (half_t)x; uint16_t b = __builtin_bit_cast(uint16_t, h);
    uint16_t o = gelu_tab[b]; y = (float)__builtin_bit_cast(half_t, o); } }
    else if (kind == 5) y = 1.0f / sqrtf(x + 1e-5f);
    else if (kind == 6) y = (float)(1.0 / (double)x);
    else y = (float)log10((double)x);
    out[i] = y;
}
extern "C" int skw_debug_math(skw_ctx* c, int kind, const float* in, float* out, long n) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    float *di = nullptr, *dout = nullptr; HIPCHK(hipMalloc((void**)&di, n * 4)); HIPCHK(hipMalloc((void**)&dout, n * 4));
    HIPCHK(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, kind, di, dout, n, c->m->gelu_tab);
    HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost)); hipFree(di); hipFree(dout); return 0;
}

// measurement hook (tools/gemm16_probe.py): one f16-MFMA GEMM of the given shape on scratch buffers, timed with HIP events on the engine stream.
// probe: bit 0 no K-loop DMA, bit 1 no MFMAs, bit 2 no epilogue stores; epi: EPI_* of skw_kernels.h (n_ctx / H / Tpad taken from the model)
extern "C" int skw_debug_gemm16(skw_ctx* c, int M, int N, int K, int epi, int probe, int iters, float* ms_per_launch) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    half_t *A = nullptr, *W = nullptr; void* C = nullptr; float *bias = nullptr, *res = nullptr;
    const size_t cbytes = (size_t)M * N * 4 + (size_t)64 * c->Tpad * N;
    // probe bits 12-19 = n (decode shapes): the launches walk n copies of W (n x N x K x 2 bytes > the 256 MB Infinity Cache: every launch finds its weights in HBM, as a decode step does)
    const int wcycle = M <= 64 ? std::max(1, (probe >> 12) & 255) : 1; if (M <= 64) probe &= 0xfff;      // (big shapes: bits 10-16 are k_gemm16w's measurement hooks)
    HIPCHK(hipMalloc((void**)&A, (size_t)M * K * 2)); HIPCHK(hipMalloc((void**)&W, (size_t)N * K * 2 * wcycle));
    HIPCHK(hipMalloc(&C, cbytes)); HIPCHK(hipMalloc((void**)&bias, (size_t)std::max(M, N) * 4)); HIPCHK(hipMalloc((void**)&res, (size_t)M * N * 4));
    { std::vector<uint16_t> h((size_t)std::max(M, N) * K); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = skw_f32_to_f16(((x >> 8) & 0xffff) / 65536.0f - 0.5f); }
      HIPCHK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
      for (int w = 0; w < wcycle; ++w) HIPCHK(hipMemcpy(W + (size_t)w * N * K, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice)); }
    HIPCHK(hipMemset(bias, 0, (size_t)std::max(M, N) * 4)); HIPCHK(hipMemset(res, 0, (size_t)M * N * 4));
    SkwGemmArgs a{}; a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = N; a.bias = bias; a.epi = epi; a.scale = 1.0f; a.probe = probe;
    a.gelu_tab = c->m->gelu_tab; a.pe = res; a.n_ctx = c->m->hp.n_audio_ctx; a.H = N / 64; a.Tpad = c->Tpad; if (epi == EPI_F32 && N < 8192) { a.res = res; a.ldres = N; }
    if (N >= 8192) a.bias = nullptr;                               // the vocabulary projection: no bias, no residual
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const bool small = M <= 64;                                      // the decode-step form: a chain of dependent launches, so the figure includes the kernel boundary
    if (small && epi == EPI_DEC_QKV) { a.C2 = res; a.C3 = res; a.ldc2 = N; a.n_ctx = N / 3; a.ldc = N / 3; }
    if (small && (probe & 64) && !(M & 15)) { a.a_frag = 1; a.probe &= ~64; }      // probe bit 6 (decode shapes, timing only): the activations addressed as a fragment-order image
    half_t* Wfrag = nullptr;      // probe bit 5 (decode shapes): the weights as fragment-order images (one per W copy of the cycle), what the step's launches read
    if (small && (probe & 32) && !(N & 15)) { HIPCHK(hipMalloc((void**)&Wfrag, (size_t)N * K * 2 * wcycle));
    for (int w = 0; w < wcycle; ++w) skw_make_wfrag(W + (size_t)w * N * K, K, N, K, epi == EPI_GELU_F16_KPERM, Wfrag + (size_t)w * N * K, c->stream);
    a.Wf = Wfrag; a.probe &= ~32; }
    // probe bit 9 (big shapes): the weights also as a fragment-order image and no measurement hooks — the launch takes k_gemm16w, as the encoder's do
    if (!small && (probe & 512) && !(N & 15)) {
        HIPCHK(hipMalloc((void**)&Wfrag, (size_t)N * K * 2));
        skw_make_wfrag(W, K, N, K, epi == EPI_GELU_F16_KPERM || epi == EPI_HEADS_F16, Wfrag, c->stream);
        a.Wf = Wfrag; a.probe = probe & ~1023;
    }
    for (int i = 0; i < 3; ++i) { if (small) skw_gemm16_small(a, c->stream); else skw_gemm16(a, c->stream); }
    HIPCHK(hipEventRecord(e0, c->stream));
    // (a launch that also touched the NEXT launch's copy of W — LDS-DMA into a scratch slab, so the bytes sit in the Infinity Cache when wanted — measured no gain:
    //  7.63 vs 7.66 us for the QKV shape.  The cold-weight cost is the transfer into the consuming XCD's L2, ~1 us per 3.5 MB whether it starts in HBM or in the Infinity Cache.)
    for (int i = 0; i < iters; ++i) { a.W = W + (size_t)(i % wcycle) * N * K;
    if (Wfrag) a.Wf = Wfrag + (size_t)(i % wcycle) * N * K; if (small) skw_gemm16_small(a, c->stream); else skw_gemm16(a, c->stream); }
    HIPCHK(hipEventRecord(e1, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); *ms_per_launch = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(A); hipFree(W); hipFree(Wfrag); hipFree(C); hipFree(bias); hipFree(res);
    return 0;
}

// tests (tests/test_gpu_gemm16w.py; the regression cover of the round-4 k_gemm16w fault): ONE product on seeded operands through k_gemm16w — the encoder's GEMM, weights from a
// fragment-order image, two memory queues counted by hand — and through k_gemm16 (both operands through LDS), every output byte compared.  epi: EPI_* of skw_kernels.h;
// frag: EPI_F16_PLAIN / EPI_VT_F16 write the fragment-order cross K / V^T image; n_ctx / Tpad: rows per clip of the per-clip layouts (M must be a multiple of n_ctx for them);
// ngroups: GEMM16W_NGROUPS for the k_gemm16w launch; with_res: EPI_F32 adds a residual.  Returns the number of differing bytes (0 = bit-identical), -2 when skw_gemm16 would not
// take k_gemm16w for this geometry (nothing compared), -1 on error.
extern "C" long skw_debug_gemm16_compare(skw_ctx* c, int M, int N, int K, int epi, int frag, int n_ctx, int Tpad, int ngroups, int with_res) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    if (M < 1 || N < 16 || (N & 15) || (K & 63) || n_ctx < 1 || (Tpad & 31) || Tpad < n_ctx) { snprintf(errbuf, 512, "skw_debug_gemm16_compare: bad geometry"); return -1; }
    const bool per_clip = epi == EPI_HEADS_F16 || epi == EPI_VT_F16 || epi == EPI_GELU_F16_KPERM_ROWPAD || epi == EPI_CONV2 || frag;
    if (per_clip && M % n_ctx) { snprintf(errbuf, 512, "skw_debug_gemm16_compare: M must be whole clips for this epilogue"); return -1; }
    if ((epi == EPI_HEADS_F16 || epi == EPI_VT_F16 || frag) && (N & 63)) { snprintf(errbuf, 512, "skw_debug_gemm16_compare: N must be whole heads for this epilogue"); return -1; }
    const size_t cbytes = ((size_t)(M / n_ctx + 2) * (Tpad + 2) * N + (size_t)M * N) * 4 + 4096;
    half_t *A = nullptr, *W = nullptr, *Wf = nullptr; char *C1 = nullptr, *C2 = nullptr; float *bias = nullptr, *res = nullptr, *pe = nullptr;
    auto cleanup = [&]() { hipFree(A); hipFree(W); hipFree(Wf); hipFree(C1); hipFree(C2); hipFree(bias); hipFree(res); hipFree(pe); };
    auto chk = [&](hipError_t e) { if (e != hipSuccess) { snprintf(errbuf, 512, "skw_debug_gemm16_compare: %s", hipGetErrorString(e)); cleanup(); return false; } return true; };
    if (!chk(hipMalloc((void**)&A, (size_t)M * K * 2)) || !chk(hipMalloc((void**)&W, (size_t)N * K * 2)) || !chk(hipMalloc((void**)&Wf, (size_t)N * K * 2)) || !chk(hipMalloc((void**)&C1, cbytes)) ||
        !chk(hipMalloc((void**)&C2, cbytes)) || !chk(hipMalloc((void**)&bias, (size_t)std::max(M, N) * 4)) || !chk(hipMalloc((void**)&res, (size_t)M * N * 4)) ||
        !chk(hipMalloc((void**)&pe, (size_t)n_ctx * N * 4))) return -1;
    {   // seeded operands: values of order 1 / sqrt(K) so that sums stay well inside f16's range after the epilogue
        uint32_t x = 0x9e3779b9u ^ (uint32_t)(M * 31 + N * 17 + K * 7 + epi);
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 65536.0f - 0.5f; };
        const float sc = 2.0f / sqrtf((float)K);
        std::vector<uint16_t> h((size_t)std::max(M, N) * K);
        for (auto& v : h) v = skw_f32_to_f16(rnd() * sc);
        if (!chk(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice))) return -1;
        for (auto& v : h) v = skw_f32_to_f16(rnd() * 2.0f);
        if (!chk(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice))) return -1;
        std::vector<float> f((size_t)std::max((size_t)M * N, (size_t)n_ctx * N));
        for (auto& v : f) v = rnd();
        if (!chk(hipMemcpy(res, f.data(), (size_t)M * N * 4, hipMemcpyHostToDevice)) || !chk(hipMemcpy(pe, f.data(), (size_t)n_ctx * N * 4, hipMemcpyHostToDevice)) ||
            !chk(hipMemcpy(bias, f.data(), (size_t)std::max(M, N) * 4, hipMemcpyHostToDevice))) return -1;
    }
    const bool perm = epi == EPI_F16_KPERM || epi == EPI_GELU_F16_KPERM || epi == EPI_GELU_F16_KPERM_ROWPAD || epi == EPI_HEADS_F16;
    skw_make_wfrag(W, K, N, K, perm ? 1 : 0, Wf, c->stream);
    SkwGemmArgs a{}; a.A = A; a.lda = K; a.W = W; a.ldw = K; a.Wf = Wf; a.M = M; a.N = N; a.K = K; a.ldc = N; a.bias = bias; a.epi = epi; a.scale = 0.125f;
    a.has_scale = (epi == EPI_F16_KPERM || epi == EPI_HEADS_F16 || epi == EPI_F16_PLAIN) ? 1 : 0;
    a.gelu_tab = c->m->gelu_tab; a.pe = pe; a.n_ctx = per_clip ? n_ctx : 0; a.H = N / 64; a.Tpad = Tpad; a.frag = frag;
    if (epi == EPI_F32 && with_res) { a.res = res; a.ldres = N; }
    if (!skw_gemm16_takes_w(a)) { cleanup(); return -2; }
    (void)skw_sw(0);
    const int ng_was = g_sw_val[SW_GEMM16W_NGROUPS], w_was = g_sw_val[SW_GEMM16W];
    if (!chk(hipMemsetAsync(C1, 0xAB, cbytes, c->stream)) || !chk(hipMemsetAsync(C2, 0xAB, cbytes, c->stream))) return -1;
    g_sw_val[SW_GEMM16W] = 1; g_sw_val[SW_GEMM16W_NGROUPS] = ngroups; a.C = C1; skw_gemm16(a, c->stream);
    g_sw_val[SW_GEMM16W] = 0; a.C = C2; skw_gemm16(a, c->stream);
    g_sw_val[SW_GEMM16W] = w_was; g_sw_val[SW_GEMM16W_NGROUPS] = ng_was;
    if (!chk(hipStreamSynchronize(c->stream)) || !chk(hipGetLastError())) return -1;
    std::vector<char> h1(cbytes), h2(cbytes);
    if (!chk(hipMemcpy(h1.data(), C1, cbytes, hipMemcpyDeviceToHost)) || !chk(hipMemcpy(h2.data(), C2, cbytes, hipMemcpyDeviceToHost))) return -1;
    long diff = 0, written = 0;
    for (size_t i = 0; i < cbytes; ++i) { diff += h1[i] != h2[i]; written += (unsigned char)h2[i] != 0xAB; }
    cleanup();
    if (written < (long)M * N) { snprintf(errbuf, 512, "skw_debug_gemm16_compare: only %ld bytes written for %d x %d outputs", written, M, N); return -1; }
    return diff;
}

// tests (tests/test_gpu_mfma_model.py): P independent v_mfma_f32_16x16x32_f16 instructions — A: P x [16][32] f16 (row i, slot k), B: P x [32][16] f16 (slot k, column j), C / D: P x [16][16] f32 —
// what the committed hardware vectors (tests/golden/mfma_f16_hw_vectors.npz) were taken with and are re-taken with on every GPU run
__global__ void k_debug_mfma16x32(const half_t* A, const half_t* B, const float* C, float* D) {
    const int p = blockIdx.x, l = threadIdx.x, r16 = l & 15, g = l >> 4;
    const half_t* a = A + (size_t)p * 512; const half_t* b = B + (size_t)p * 512; const float* c = C + (size_t)p * 256; float* d = D + (size_t)p * 256;
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = a[r16 * 32 + 8 * g + e]; fb[e] = b[(8 * g + e) * 16 + r16]; }      // lane (row / column r16, group g) holds slots 8 g .. 8 g + 7
    f32x4 acc; for (int r = 0; r < 4; ++r) acc[r] = c[(4 * g + r) * 16 + r16];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + r16] = acc[r];
}
extern "C" int skw_debug_mfma16x32(skw_ctx* c, long P, const uint16_t* A_host, const uint16_t* B_host, const float* C_host, float* D_host) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    if (P < 1 || P > (1 << 20)) { snprintf(errbuf, 512, "skw_debug_mfma16x32: bad problem count"); return -1; }
    char* buf = nullptr;
    auto chk = [&](hipError_t e) { if (e != hipSuccess) { snprintf(errbuf, 512, "skw_debug_mfma16x32: %s", hipGetErrorString(e)); hipFree(buf); return false; } return true; };
    if (!chk(hipMalloc((void**)&buf, (size_t)P * 4096))) return -1;
    half_t* A = (half_t*)buf; half_t* B = (half_t*)(buf + P * 1024); float* C = (float*)(buf + P * 2048); float* D = (float*)(buf + P * 3072);
    if (!chk(hipMemcpy(A, A_host, P * 1024, hipMemcpyHostToDevice)) || !chk(hipMemcpy(B, B_host, P * 1024, hipMemcpyHostToDevice))) return -1;
    if (!chk(hipMemcpy(C, C_host, P * 1024, hipMemcpyHostToDevice))) return -1;
    hipLaunchKernelGGL(k_debug_mfma16x32, dim3((unsigned)P), dim3(64), 0, c->stream, A, B, C, D);
    if (!chk(hipStreamSynchronize(c->stream)) || !chk(hipGetLastError()) || !chk(hipMemcpy(D_host, D, P * 1024, hipMemcpyDeviceToHost))) return -1;
    hipFree(buf);
    return 0;
}
// tests (tests/test_gpu_mfma_model.py): C = A . W^T through one of the f16_mfma GEMM kernels, plain f32 epilogue (no bias, no residual), operands given as f16 bit patterns in the
// kernels' memory order ([M][K] and [N][K], K axis as the kernels load it) — to be compared bit for bit with oracle/'s restatement of the matrix cores (skwo_gemm_f16mfma).
// kernel: 0 = k_gemm16w (weights from a fragment-order image), 1 = k_gemm16 (both operands through LDS), 2 = the decode step's product: k_gemm16_small (four waves split K) for N < 8192,
// k_gemm16_vocab (one wave chains the whole K) from there on
extern "C" int skw_debug_gemm16_out(skw_ctx* c, int kernel, int M, int N, int K, const uint16_t* A_host, const uint16_t* W_host, float* C_host) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    if (M < 1 || (N & 15) || (K & 127) || kernel < 0 || kernel > 2) { snprintf(errbuf, 512, "skw_debug_gemm16_out: bad geometry"); return -1; }
    half_t *A = nullptr, *W = nullptr, *Wf = nullptr; float* C = nullptr;
    auto cleanup = [&]() { hipFree(A); hipFree(W); hipFree(Wf); hipFree(C); };
    auto chk = [&](hipError_t e) { if (e != hipSuccess) { snprintf(errbuf, 512, "skw_debug_gemm16_out: %s", hipGetErrorString(e)); cleanup(); return false; } return true; };
    if (!chk(hipMalloc((void**)&A, (size_t)M * K * 2)) || !chk(hipMalloc((void**)&W, (size_t)N * K * 2)) || !chk(hipMalloc((void**)&Wf, (size_t)N * K * 2))) return -1;
    if (!chk(hipMalloc((void**)&C, (size_t)M * N * 4)) || !chk(hipMemset(C, 0xAB, (size_t)M * N * 4))) return -1;
    if (!chk(hipMemcpy(A, A_host, (size_t)M * K * 2, hipMemcpyHostToDevice)) || !chk(hipMemcpy(W, W_host, (size_t)N * K * 2, hipMemcpyHostToDevice))) return -1;
    skw_make_wfrag(W, K, N, K, 0, Wf, c->stream);
    SkwGemmArgs a{}; a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = N; a.epi = EPI_F32; a.scale = 1.0f;
    (void)skw_sw(0);
    const int w_was = g_sw_val[SW_GEMM16W];
    if (kernel == 2) { a.Wf = Wf; if (!skw_gemm16_small(a, c->stream)) { snprintf(errbuf, 512, "skw_debug_gemm16_out: k_gemm16_small does not take this geometry"); cleanup(); return -1; } }
    else {
        a.Wf = kernel == 0 ? Wf : nullptr; g_sw_val[SW_GEMM16W] = kernel == 0 ? 1 : 0;
        if (kernel == 0 && !skw_gemm16_takes_w(a)) { g_sw_val[SW_GEMM16W] = w_was; snprintf(errbuf, 512, "skw_debug_gemm16_out: k_gemm16w does not take this geometry"); cleanup(); return -1; }
        skw_gemm16(a, c->stream); g_sw_val[SW_GEMM16W] = w_was;
    }
    if (!chk(hipStreamSynchronize(c->stream)) || !chk(hipGetLastError()) || !chk(hipMemcpy(C_host, C, (size_t)M * N * 4, hipMemcpyDeviceToHost))) return -1;
    cleanup();
    return 0;
}

// the launch clock's records (SkwKClk, skw_kernels.h): one clock per (row group, decoder layer) graph node
static const int KCLK_CAP = 1024;                                       // launches recorded per node and call (a 30 s window is <= 466 steps)
static size_t kclk_node_bytes() { return sizeof(SkwKClk) + sizeof(SkwKClkRec) * SKW_KCLK_SHARDS * (KCLK_CAP - 1); }
// one launch's record: the maxima over its shards; false when no live workgroup stamped it
static bool kclk_launch(const SkwKClk* k, int j, unsigned long long* t0, unsigned long long* t1, unsigned* live) {
    unsigned long long a = 0, b = 0; unsigned l = 0;
    for (int s = 0; s < SKW_KCLK_SHARDS; ++s) { const SkwKClkRec& r = k->rec[j][s]; a = std::max(a, r.t0_inv); b = std::max(b, r.t1); l += r.live_rows; }
    if (!a || !b) return false;
    *t0 = ~a; *t1 = b; *live = l; return true;
}
static SkwKClk* kclk_node(const skw_ctx* c, int g, int l) {
    return c->kclk_on ? (SkwKClk*)((char*)c->kclk + kclk_node_bytes() * ((size_t)g * c->m->hp.n_text_layer + l)) : nullptr;
}
static int kclk_reset(skw_ctx* c) {
    char* errbuf = c->errbuf;
    const size_t nb = kclk_node_bytes(), n_nodes = (size_t)skw_ctx::MAX_GROUPS * c->m->hp.n_text_layer;
    HIPCHK(hipMemsetAsync(c->kclk, 0, nb * n_nodes, c->stream));
    static const unsigned hdr[4] = {(unsigned)KCLK_CAP, 0, 0, 0};
    for (size_t i = 0; i < n_nodes; ++i) HIPCHK(hipMemcpyAsync((char*)c->kclk + nb * i, hdr, 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
// Arms / disarms the in-kernel launch clock of the decode step's cross attention (f16_mfma, fragment-order images).  Armed, the step graphs are captured with the clock's
// pointers (their key carries the flag) and every skw_full_batch starts from zeroed records; skw_ctx_kernel_clock_get sums what the last call recorded.
extern "C" int skw_ctx_kernel_clock(skw_ctx* c, int on) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    if (on && !c->kclk) {
        HIPCHK(hipMalloc((void**)&c->kclk, kclk_node_bytes() * skw_ctx::MAX_GROUPS * c->m->hp.n_text_layer));
        int khz = 0; HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->m->device));
        if (khz <= 0) { snprintf(errbuf, 512, "device reports no wall clock rate"); return -1; }
        c->kclk_khz = khz;
    }
    c->kclk_on = on ? 1 : 0;
    return on ? kclk_reset(c) : 0;
}
// launches recorded by the last skw_full_batch: count, summed first-wave-in -> last-wave-out microseconds, summed live rows (the launch's algorithmic bytes are
// 4 B x live rows x n_audio_ctx x n_text_state), shortest and longest launch
extern "C" int skw_ctx_kernel_clock_get(skw_ctx* c, long* launches, double* sum_us, double* sum_live_rows, double* min_us, double* max_us, int* clock_khz) {
    char* errbuf = c->errbuf;
    if (!c->kclk) { snprintf(errbuf, 512, "the kernel clock was never armed"); return -1; }
    HIPCHK(hipSetDevice(c->m->device));
    const size_t nb = kclk_node_bytes(), n_nodes = (size_t)skw_ctx::MAX_GROUPS * c->m->hp.n_text_layer;
    std::vector<char> h(nb * n_nodes);
    HIPCHK(hipMemcpy(h.data(), c->kclk, h.size(), hipMemcpyDeviceToHost));
    long n = 0; double su = 0, sl = 0, mn = 1e30, mx = 0;
    for (size_t i = 0; i < n_nodes; ++i) {
        const SkwKClk* k = (const SkwKClk*)(h.data() + nb * i);
        for (int j = 0; j < KCLK_CAP; ++j) {
            unsigned long long t0, t1; unsigned live;
            if (!kclk_launch(k, j, &t0, &t1, &live)) continue;
            const double us = (double)(t1 - t0) * 1000.0 / c->kclk_khz;
            ++n; su += us; sl += live; mn = std::min(mn, us); mx = std::max(mx, us);
        }
    }
    *launches = n; *sum_us = su; *sum_live_rows = sl; *min_us = n ? mn : 0; *max_us = mx; *clock_khz = c->kclk_khz;
    return 0;
}

// every recorded launch of the last call as (begin us, end us, live rows), times relative to the earliest begin; returns the count written (<= cap), < 0 on error.  Launches of
// different row groups overlap in time: the union of the intervals is the time the kernel was in flight at all.
extern "C" long skw_ctx_kernel_clock_records(skw_ctx* c, double* out /* [cap][3] */, long cap) {
    char* errbuf = c->errbuf;
    if (!c->kclk) { snprintf(errbuf, 512, "the kernel clock was never armed"); return -1; }
    HIPCHK(hipSetDevice(c->m->device));
    const size_t nb = kclk_node_bytes(), n_nodes = (size_t)skw_ctx::MAX_GROUPS * c->m->hp.n_text_layer;
    std::vector<char> h(nb * n_nodes);
    HIPCHK(hipMemcpy(h.data(), c->kclk, h.size(), hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull;
    unsigned long long t0, t1; unsigned live;
    for (size_t i = 0; i < n_nodes; ++i) { const SkwKClk* k = (const SkwKClk*)(h.data() + nb * i);
        for (int j = 0; j < KCLK_CAP; ++j) if (kclk_launch(k, j, &t0, &t1, &live)) t_min = std::min(t_min, t0); }
    long n = 0;
    for (size_t i = 0; i < n_nodes; ++i) { const SkwKClk* k = (const SkwKClk*)(h.data() + nb * i);
        for (int j = 0; j < KCLK_CAP && n < cap; ++j) { if (!kclk_launch(k, j, &t0, &t1, &live)) continue;
            out[3 * n] = (double)(t0 - t_min) * 1000.0 / c->kclk_khz; out[3 * n + 1] = (double)(t1 - t_min) * 1000.0 / c->kclk_khz; out[3 * n + 2] = live; ++n; } }
    return n;
}

// The decode step's cross attention alone, B rows, launched back to back over `layers` different K / V^T images (so no launch re-reads what a previous one left in a cache).
// Every launch is stamped twice: by HIP events at the kernel's own begin and end (hipExtLaunchKernelGGL: what rocprofv3 reports as the duration) and — the one-pass kernel — by its
// in-kernel clock (first wave in to last wave out).  us_per_launch: the event average; us_per_launch_clock (may be null): the clock's average, 0 where the kernel has no clock.
extern "C" int skw_debug_xattn(skw_ctx* c, int B, int layers, int probe, int iters, float* us_per_launch, float* us_per_launch_clock) {
    char* errbuf = c->errbuf; HIPCHK(hipSetDevice(c->m->device));
    if (probe) { snprintf(errbuf, 512, "skw_debug_xattn: the parts-off probe launches were removed with round 5's pruning (results: profiles/r02f, r03b)"); return -1; }
    const skw_hparams& hp = c->m->hp; const int d = hp.n_text_state, H = hp.n_text_head, nc = hp.n_audio_ctx, Tpad = c->Tpad;
    const size_t kn = (size_t)B * Tpad * d, vn = (size_t)B * H * 64 * Tpad;
    half_t *K = nullptr, *V = nullptr, *q = nullptr, *out = nullptr; SkwKClk* clk = nullptr;
    iters = std::max(1, std::min(iters, KCLK_CAP));
    HIPCHK(hipMalloc((void**)&K, kn * 2 * layers)); HIPCHK(hipMalloc((void**)&V, vn * 2 * layers));
    HIPCHK(hipMalloc((void**)&q, (size_t)B * d * 2)); HIPCHK(hipMalloc((void**)&out, (size_t)B * d * 4));
    HIPCHK(hipMalloc((void**)&clk, kclk_node_bytes()));
    { std::vector<uint16_t> h(std::max(kn, vn)); uint32_t x = 777; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = skw_f32_to_f16((((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.25f); }
      for (int l = 0; l < layers; ++l) { HIPCHK(hipMemcpy(K + kn * l, h.data(), kn * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(V + vn * l, h.data(), vn * 2, hipMemcpyHostToDevice)); }
      HIPCHK(hipMemcpy(q, h.data(), (size_t)B * d * 2, hipMemcpyHostToDevice)); }
    const int pv16 = c->kv_frag() ? 2 : c->precision == SKW_PRECISION_F16_MFMA;       // (timing only: the images are random bytes in either layout; Tpad * d elements per slot are allocated above)
    for (int i = 0; i < layers; ++i) skw_dec_cross_attn_vt(q, K + kn * (i % layers), V + vn * (i % layers), B, H, d, nc, Tpad, out, nullptr, c->stream, 0, pv16);
    HIPCHK(hipMemsetAsync(clk, 0, kclk_node_bytes(), c->stream));
    { static const unsigned hdr[4] = {(unsigned)KCLK_CAP, 0, 0, 0}; HIPCHK(hipMemcpyAsync(clk, hdr, 16, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    float ms = 0;
    std::vector<hipEvent_t> ev(2 * (size_t)iters);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    for (int i = 0; i < iters; ++i)
        skw_dec_cross_attn_vt(q, K + kn * (i % layers), V + vn * (i % layers), B, H, d, nc, Tpad, out, nullptr, c->stream, 0, pv16, nullptr, ev[2 * i], ev[2 * i + 1], 0, pv16 == 2 ? clk : nullptr);
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < iters; ++i) { float t = 0; if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) == hipSuccess) ms += t; }
    for (auto& e : ev) hipEventDestroy(e);
    *us_per_launch = 1000.0f * ms / iters;
    if (us_per_launch_clock) {
        *us_per_launch_clock = 0.0f;
        if (pv16 == 2) {
            int khz = 0; HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->m->device));
            std::vector<char> h(kclk_node_bytes()); HIPCHK(hipMemcpy(h.data(), clk, h.size(), hipMemcpyDeviceToHost));
            const SkwKClk* k = (const SkwKClk*)h.data(); double su = 0; int n = 0; unsigned long long t0, t1; unsigned live;
            for (int j = 0; j < iters; ++j) if (kclk_launch(k, j, &t0, &t1, &live)) { su += (double)(t1 - t0) * 1000.0 / std::max(1, khz); ++n; }
            if (n) *us_per_launch_clock = (float)(su / n);
        }
    }
    hipFree(K); hipFree(V); hipFree(q); hipFree(out); hipFree(clk);
    return 0;
}

// ------------------------------------------------------------------ resampler front end (R1-R3), model-free device context
struct skw_dsp { int device = 0; hipStream_t stream = nullptr;
char errbuf[512] = {0}; float *d_in = nullptr, *d_out = nullptr, *d_frac = nullptr, *d_coef = nullptr;
int* d_pos = nullptr; int* d_n = nullptr; double* d_li = nullptr;
                 size_t cap_in = 0, cap_out = 0; int coef_L = 0, coef_M = 0;
                 double* d_start = nullptr; int *d_count = nullptr, *d_offset = nullptr, *d_flag = nullptr; size_t cap_chunks = 0; int last_flags[2] = {0, 0}; bool host_walk_last = false; };
extern "C" skw_dsp* skw_dsp_create(int device, char* err, size_t errlen) {
    int ndev = skw_device_count();
    if (ndev <= 0) { set_err(err, errlen, "no HIP device available: the resampler kernels require an MI355X (gfx950); there is no CPU fallback"); return nullptr; }
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "gpu_device %d out of range (%d devices)", device, ndev); return nullptr; }
    skw_dsp* d = new skw_dsp(); d->device = device;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&d->d_n, sizeof(int)) != hipSuccess || hipMalloc((void**)&d->d_li,
        sizeof(double)) != hipSuccess) { set_err(err, errlen, "device allocation failed");
    delete d; return nullptr; }
    return d;
}
extern "C" void skw_dsp_free(skw_dsp* d) { if (!d) return; hipSetDevice(d->device); hipStreamSynchronize(d->stream); hipFree(d->d_in); hipFree(d->d_out);
hipFree(d->d_frac); hipFree(d->d_pos); hipFree(d->d_coef); hipFree(d->d_n); hipFree(d->d_li); hipFree(d->d_start); hipFree(d->d_count); hipFree(d->d_offset);
hipFree(d->d_flag); hipStreamDestroy(d->stream); delete d; }
extern "C" const char* skw_dsp_last_error(const skw_dsp* d) { return d->errbuf; }
static int dsp_reserve(skw_dsp* d, size_t n_in, size_t n_out) {
    char* errbuf = d->errbuf;
    if (n_in > d->cap_in) {
        hipFree(d->d_in); d->cap_in = n_in * 2; HIPCHK(hipMalloc((void**)&d->d_in, d->cap_in * sizeof(float))); poison_floats(d->d_in, d->cap_in * sizeof(float));
    }
    if (n_out > d->cap_out) { hipFree(d->d_out); hipFree(d->d_frac);
    hipFree(d->d_pos); d->cap_out = n_out * 2; HIPCHK(hipMalloc((void**)&d->d_out, d->cap_out * sizeof(float)));
    HIPCHK(hipMalloc((void**)&d->d_frac, d->cap_out * sizeof(float))); HIPCHK(hipMalloc((void**)&d->d_pos, d->cap_out * sizeof(int)));
    poison_floats(d->d_out, d->cap_out * sizeof(float)); poison_floats(d->d_frac, d->cap_out * sizeof(float)); }
    return 0;
}
extern "C" void skw_resampler_init(skw_resampler_state* st, double ratio, int chunk_frames, int channels) {
    memset(st, 0, sizeof *st); st->last_index = -4.0; st->ratio = ratio; st->chunk_frames = chunk_frames; st->channels = channels;   // -(POLYNOMIAL_LEN/2), zero history
}
// n_chunks full chunks of interleaved input -> interleaved output frames; state (history + fractional index) carried like rubato's
extern "C" int skw_resample_linear(skw_dsp* d, skw_resampler_state* st, const float* in, int n_chunks, float* out, int out_cap_frames, int* out_frames) {
    char* errbuf = d->errbuf; HIPCHK(hipSetDevice(d->device));
    const int ch = st->channels, chunk = st->chunk_frames; *out_frames = 0;
    if (ch < 1 || ch > 2 || chunk < 1 || n_chunks < 1 || (chunk < 16 && n_chunks != 1)) { snprintf(errbuf, 512, "resampler: unsupported geometry (channels %d, chunk_frames %d)", ch, chunk);
    return -1; }
    const size_t n_in = (size_t)(16 + (size_t)n_chunks * chunk) * ch;
    if (dsp_reserve(d, n_in, (size_t)out_cap_frames * ch)) return -1;
    HIPCHK(hipMemcpyAsync(d->d_in, st->hist, sizeof(float) * 16 * ch, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_in + 16 * ch, in, sizeof(float) * (size_t)n_chunks * chunk * ch, hipMemcpyHostToDevice, d->stream));
    if ((size_t)n_chunks + 1 > d->cap_chunks) {
        hipFree(d->d_start); hipFree(d->d_count); hipFree(d->d_offset); d->cap_chunks = ((size_t)n_chunks + 1) * 2;
        HIPCHK(hipMalloc((void**)&d->d_start, d->cap_chunks * sizeof(double))); HIPCHK(hipMalloc((void**)&d->d_count, d->cap_chunks * sizeof(int)));
        HIPCHK(hipMalloc((void**)&d->d_offset, d->cap_chunks * sizeof(int)));
        if (!d->d_flag) HIPCHK(hipMalloc((void**)&d->d_flag, 2 * sizeof(int)));
    }
    // The chunk starts are one sequential f64 recurrence over the whole call (rubato's idx += t_ratio).  When every addition of it is exact (48 / 32 / 96 / 8 kHz
    // sources: t_ratio has a handful of mantissa bits) a closed form gives them on the device.  When it is not (the 44.1 kHz family) and the call is long, the
    // recurrence is what a CPU core is for — ~480 k dependent additions per 30 s, 0.6 ms at four cycles each, against 7.5 ms for one GPU lane stepping binades —
    // so the host walks it, the device proves every chunk against it in parallel (k_resample_walk) and does the data path.  Short calls (a streaming packet or
    // two) keep the device's own proposals.  Same IEEE additions either way: the result is rubato's, bit for bit.
    const double t_ratio = 1.0 / st->ratio;
    const bool t_exact = chunk <= 4096 && ldexp(t_ratio, 36) == floor(ldexp(t_ratio, 36));
    const bool host_walk = !t_exact && n_chunks >= 8 && !skw_sw(SW_RESAMPLE_NO_HOST_WALK);
    if (host_walk) {
        std::vector<double> hs((size_t)n_chunks + 1); std::vector<int> hc((size_t)n_chunks + 1, 0), ho((size_t)n_chunks + 1);
        const double end_idx = (double)(chunk - 9) - ceil(t_ratio); double s = st->last_index; int off = 0;
        for (int cix = 0; cix < n_chunks; ++cix) { double x = s; int n = 0; while (x < end_idx) { x += t_ratio; n++; } hs[cix] = s; hc[cix] = n; ho[cix] = off; off += n; s = x - (double)chunk; }
        hs[n_chunks] = s; ho[n_chunks] = off;
        HIPCHK(hipMemcpyAsync(d->d_start, hs.data(), sizeof(double) * hs.size(), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemcpyAsync(d->d_count, hc.data(), sizeof(int) * hc.size(), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemcpyAsync(d->d_offset, ho.data(), sizeof(int) * ho.size(), hipMemcpyHostToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));      // (the vectors are locals)
    }
    d->host_walk_last = host_walk;
    skw_resample_linear_launch(d->d_in, ch, st->last_index, t_ratio, chunk, n_chunks, d->d_pos, d->d_frac, d->d_n, d->d_li, d->d_out, out_cap_frames,
                               d->d_start, d->d_count, d->d_offset, d->d_flag, d->stream, host_walk);
    int n = 0; double li = 0;
    HIPCHK(hipMemcpyAsync(&n, d->d_n, sizeof(int), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipMemcpyAsync(&li, d->d_li, sizeof(double), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemcpyAsync(d->last_flags, d->d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (n > out_cap_frames) { snprintf(errbuf, 512, "resampler: output capacity %d too small for %d frames", out_cap_frames, n); return -1; }
    HIPCHK(hipMemcpyAsync(out, d->d_out, sizeof(float) * (size_t)n * ch, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    {   // buffer.copy_within(chunk.., 0): the 16 frames that precede the next chunk (a chunk shorter than 16 frames shifts the old history)
        const size_t fresh = (size_t)n_chunks * chunk;
        if (fresh >= 16) memcpy(st->hist, in + (fresh - 16) * ch, sizeof(float) * 16 * ch);
        else { memmove(st->hist, st->hist + fresh * ch, sizeof(float) * (16 - fresh) * ch); memcpy(st->hist + (16 - fresh) * ch, in, sizeof(float) * fresh * ch); }
    }
    st->last_index = li; *out_frames = n; return 0;
}
// quality mode: Kaiser-windowed sinc, 32 taps x decimation factor per phase (mono or interleaved stereo)
static int dsp_polyphase_coefs(skw_dsp* d, int in_rate, int out_rate, int* L_o, int* M_o, int* T_o) {
    char* errbuf = d->errbuf;
    auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
    const long g = gcd(in_rate, out_rate); const int L = (int)(out_rate / g), M = (int)(in_rate / g);
    if (L > 4096 || M > 4096) { snprintf(errbuf, 512, "resampler: ratio %d/%d needs %d phases / a decimation of %d (> 4096)", out_rate, in_rate, L, M); return -1; }
    const int T = 32 * std::max(1, (M + L - 1) / L);     // span 32 samples of the slower rate
    *L_o = L; *M_o = M; *T_o = T;
    if (d->coef_L == L && d->coef_M == M) return 0;
    std::vector<float> h((size_t)L * T);
    const double fc = 0.5 * std::min(1.0, (double)L / M) * 0.90, beta = 8.6;   // cutoff (cycles per input sample), a little below Nyquist of the narrower side
    auto bessel0 = [](double x) { double s = 1, t = 1; for (int k = 1; k < 40; ++k) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; } return s; };
    for (int ph = 0; ph < L; ++ph) {
        double sum = 0; std::vector<double> row(T);
        for (int t = 0; t < T; ++t) {
            const double xpos = (double)(t - (T / 2 - 1)) - (double)ph / L;     // distance (in input samples) from the output instant
            const double w = std::fabs(xpos) >= T / 2 ? 0.0 : bessel0(beta * std::sqrt(1.0 - (xpos / (T / 2)) * (xpos / (T / 2)))) / bessel0(beta);
            const double arg = 2.0 * M_PI * fc * xpos; const double sinc = std::fabs(arg) < 1e-12 ? 1.0 : std::sin(arg) / arg;
            row[t] = 2.0 * fc * sinc * w; sum += row[t];
        }
        for (int t = 0; t < T; ++t) h[(size_t)ph * T + t] = (float)(row[t] / sum);   // unity DC gain per phase
    }
    hipFree(d->d_coef); d->d_coef = nullptr; HIPCHK(hipMalloc((void**)&d->d_coef, h.size() * sizeof(float))); HIPCHK(hipMemcpy(d->d_coef, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    d->coef_L = L; d->coef_M = M; return 0;
}
// whole buffer at once
extern "C" int skw_resample_polyphase(skw_dsp* d, const float* in, long n_in_frames, int channels, int in_rate, int out_rate, float* out, long out_cap_frames, long* out_frames) {
    char* errbuf = d->errbuf; HIPCHK(hipSetDevice(d->device));
    int L, M, T; if (dsp_polyphase_coefs(d, in_rate, out_rate, &L, &M, &T)) return -1;
    if (channels < 1 || channels > 2) { snprintf(errbuf, 512, "resampler: unsupported channel count %d", channels); return -1; }
    const long n_out = (n_in_frames * L + M - 1) / M;
    if (n_out > out_cap_frames) { snprintf(errbuf, 512, "resampler: output capacity too small"); return -1; }
    if (dsp_reserve(d, (size_t)n_in_frames * channels, (size_t)n_out * channels)) return -1;
    HIPCHK(hipMemcpyAsync(d->d_in, in, sizeof(float) * (size_t)n_in_frames * channels, hipMemcpyHostToDevice, d->stream));
    skw_resample_polyphase_launch(d->d_in, 0, n_in_frames, n_in_frames, channels, d->d_coef, L, M, T, d->d_out, 0, n_out, d->stream);
    HIPCHK(hipGetLastError());      // a refused launch (LDS request) must surface as an error, not as stale output
    HIPCHK(hipMemcpyAsync(out, d->d_out, sizeof(float) * (size_t)n_out * channels, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    *out_frames = n_out; return 0;
}
extern "C" int skw_dsp_last_scan_fallback(const skw_dsp* d) { return d->last_flags[1] ? 2 : ((d->last_flags[0] || d->host_walk_last) ? 1 : 0); }

// streaming polyphase: the input tail the later outputs still need stays on the device; a push uploads only the new frames and
// computes only the outputs whose filter support has arrived (all remaining ones when `final`).  Identical to the whole-buffer result.
struct skw_pp_stream { skw_dsp* d; int ch, in_rate, out_rate, L, M, T; float* d_buf = nullptr; size_t cap_frames = 0; long base = 0, total = 0, next = 0; float* d_o = nullptr; size_t cap_o = 0; };
extern "C" skw_pp_stream* skw_polyphase_stream_create(skw_dsp* d, int channels, int in_rate, int out_rate) {
    if (hipSetDevice(d->device) != hipSuccess) return nullptr;
    int L, M, T; if (channels < 1 || channels > 2) { snprintf(d->errbuf, 512, "resampler: unsupported channel count %d", channels); return nullptr; }
    if (dsp_polyphase_coefs(d, in_rate, out_rate, &L, &M, &T)) return nullptr;
    skw_pp_stream* p = new skw_pp_stream(); p->d = d; p->ch = channels; p->in_rate = in_rate; p->out_rate = out_rate; p->L = L; p->M = M; p->T = T; return p;
}
extern "C" void skw_polyphase_stream_free(skw_pp_stream* p) { if (!p) return; hipSetDevice(p->d->device); hipStreamSynchronize(p->d->stream); hipFree(p->d_buf); hipFree(p->d_o); delete p; }
extern "C" int skw_polyphase_stream_push(skw_pp_stream* p, const float* in, long n_frames, int final_call, float* out, long out_cap_frames, long* out_frames) {
    skw_dsp* d = p->d; char* errbuf = d->errbuf; HIPCHK(hipSetDevice(d->device)); *out_frames = 0;
    int L, M, T; if (dsp_polyphase_coefs(d, p->in_rate, p->out_rate, &L, &M, &T)) return -1;       // the context's table may have served another ratio since
    const long ch = p->ch;
    if (n_frames > 0) {
        const size_t have = (size_t)(p->total - p->base), need = have + (size_t)n_frames;
        if (need > p->cap_frames) {       // grow (rare: the buffer holds one packet + the retained tail)
            float* nb = nullptr; const size_t cap = need * 2 + 4096; HIPCHK(hipMalloc((void**)&nb, cap * ch * sizeof(float)));
            if (have) HIPCHK(hipMemcpyAsync(nb, p->d_buf, have * ch * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream)); hipFree(p->d_buf); p->d_buf = nb; p->cap_frames = cap;
        }
        HIPCHK(hipMemcpyAsync(p->d_buf + have * ch, in, (size_t)n_frames * ch * sizeof(float), hipMemcpyHostToDevice, d->stream));
        p->total += n_frames;
    }
    // output m sits at input position floor(m*M/L) and reads frames [pos - (T/2 - 1), pos + T/2]
    long m_hi;
    if (final_call) m_hi = (p->total * L + M - 1) / M;
    else { const long last_pos = p->total - 1 - T / 2; m_hi = last_pos < 0 ? 0 : ((last_pos + 1) * L + M - 1) / M; }
    const long n_new = m_hi - p->next;
    if (n_new > 0) {
        if (n_new > out_cap_frames) { snprintf(errbuf, 512, "resampler: output capacity too small"); return -1; }
        if ((size_t)n_new > p->cap_o) { hipFree(p->d_o); p->cap_o = (size_t)n_new * 2; HIPCHK(hipMalloc((void**)&p->d_o, p->cap_o * ch * sizeof(float))); }
        skw_resample_polyphase_launch(p->d_buf, p->base, p->total - p->base, final_call ? p->total : p->total, (int)ch, d->d_coef, L, M, T, p->d_o, p->next, n_new, d->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, p->d_o, (size_t)n_new * ch * sizeof(float), hipMemcpyDeviceToHost, d->stream));
        p->next = m_hi; *out_frames = n_new;
    }
    // drop what no later output reads: keep from the support start of output `next`
    long keep_from = (p->next * M) / L - (T / 2 - 1); if (keep_from < p->base) keep_from = p->base; if (keep_from > p->total) keep_from = p->total;
    if (keep_from > p->base) {
        const size_t keep = (size_t)(p->total - keep_from);
        if (keep) {   // overlapping move on one stream: stage through the output scratch when the ranges overlap
            if ((size_t)(keep_from - p->base) >= keep) HIPCHK(hipMemcpyAsync(p->d_buf, p->d_buf + (size_t)(keep_from - p->base) * ch, keep * ch * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
            else {
                float* tmp = nullptr; HIPCHK(hipMalloc((void**)&tmp, keep * ch * sizeof(float)));
                HIPCHK(hipMemcpyAsync(tmp, p->d_buf + (size_t)(keep_from - p->base) * ch, keep * ch * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
                HIPCHK(hipMemcpyAsync(p->d_buf, tmp, keep * ch * sizeof(float), hipMemcpyDeviceToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); hipFree(tmp);
            }
        }
        p->base = keep_from;
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}
