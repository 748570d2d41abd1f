/*
 * make_synth_model — writes a Whisper model file in whisper.cpp's legacy GGML
 * container ("ggml" magic, 11 hparams, mel filters, vocab, named tensors), the
 * format the reference plugin's `model_path` points at
 * (/root/reference/plugins/native/whisper/src/lib.rs:68-70, 360; models listed in
 * /root/reference/plugins/native/whisper/README.md:101-129).
 *
 * No trained weights exist offline, so tensors come from a counter-based
 * generator (splitmix64 keyed by seed + FNV-1a(tensor name) + element index).
 * Encoder and decoder blocks use U(-1/sqrt(fan_in), 1/sqrt(fan_in)); the three
 * decoder tensors that shape greedy decoding (token embedding, positional
 * embedding, final LayerNorm gain) are *engineered* so that a random-weight
 * model behaves like a trained one under whisper.cpp's decoding rules:
 * peaked next-token distributions (avg log-prob > -1, no temperature fallback),
 * timestamp tokens that advance with position, and an EOT that becomes likely
 * after ~100 tokens.  Which text token / which timestamp wins still depends on
 * the audio through the full encoder/decoder stack, so parity tests on the
 * emitted token ids exercise every kernel.  See DESIGN.md §"Synthetic model".
 *
 * usage: make_synth_model OUT.bin [--size tiny|base|small] [--vocab 51864|51865|51866] [--mels 80|128] [--seed N] [--f32]
 *                         [--key value ...]   (see `knobs` below)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/skw_math.h"

typedef struct {
    int n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
} hparams_t;

/* engineered-decoder knobs (all overridable from the command line) */
static double k_emb_scale = 0.05;   /* text-token embedding amplitude (U(-s,s)) */
static double k_gamma_text = 30.0;  /* final LN gain on text dims */
static double k_gamma_ctl = 200.0;   /* final LN gain on control dims */
static double k_pos_scale = 0.6;    /* positional embedding random part */
static double k_ts_rho = 0.5;       /* timestamp tokens: scale of random part */
static double k_eot_lo = -6.0, k_eot_hi = 4.0; /* EOT ramp on control dim c0 */
static double k_eot_p0 = 60.0, k_eot_p1 = 200.0;
static double k_ts_on = 5.0, k_ts_off = -5.0;  /* timestamp-class pulse on c1 */
static double k_ts_period = 12.0;
static double k_ts_speed = 15.5;    /* target timestamp index per position */
static double k_tc_amp = 0.6;       /* time-code amplitude in P */
static double k_ctl_w = 1.0;        /* amplitude of control components in E */
static double k_enc_gain = 1.0;     /* encoder weight gain */
static double k_conv1_gain = 10.0, k_conv2_gain = 3.0; /* make the audio, not the positional embedding, drive the encoder */
static double k_enc_mlp_gain = 4.0;
static double k_dec_fc1_gain = 4.0, k_dec_fc2_gain = 2.0; /* chaotic (hash-like) decoder MLPs */
static double k_dec_qk_gain = 3.0, k_dec_vo_gain = 0.5, k_cross_qk_gain = 4.0, k_cross_vo_gain = 0.7;

typedef struct { const char* name; double* v; } knob_t;
static knob_t knobs[] = {
    {"emb_scale", &k_emb_scale}, {"gamma_text", &k_gamma_text}, {"gamma_ctl", &k_gamma_ctl},
    {"pos_scale", &k_pos_scale}, {"ts_rho", &k_ts_rho}, {"eot_lo", &k_eot_lo}, {"eot_hi", &k_eot_hi},
    {"eot_p0", &k_eot_p0}, {"eot_p1", &k_eot_p1}, {"ts_on", &k_ts_on}, {"ts_off", &k_ts_off},
    {"ts_period", &k_ts_period}, {"ts_speed", &k_ts_speed}, {"tc_amp", &k_tc_amp}, {"ctl_w", &k_ctl_w},
    {"enc_gain", &k_enc_gain}, {"conv1_gain", &k_conv1_gain}, {"conv2_gain", &k_conv2_gain}, {"enc_mlp_gain", &k_enc_mlp_gain},
    {"dec_fc1_gain", &k_dec_fc1_gain}, {"dec_fc2_gain", &k_dec_fc2_gain}, {"dec_qk_gain", &k_dec_qk_gain}, {"dec_vo_gain", &k_dec_vo_gain},
    {"cross_qk_gain", &k_cross_qk_gain}, {"cross_vo_gain", &k_cross_vo_gain}, {NULL, NULL}};

static uint64_t g_seed = 1234;
static int g_f16 = 1;
static FILE* g_out;

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t fnv1a(const char* s) { uint64_t h = 1469598103934665603ULL; while (*s) { h ^= (unsigned char)*s++; h *= 1099511628211ULL; } return h; }
/* uniform in (-1,1), 24 bits */
static float urand(uint64_t key, uint64_t idx) {
    uint64_t z = mix64(key + (idx + 1) * 0x9E3779B97F4A7C15ULL);
    return (float)((double)(z >> 40) * (1.0 / 8388608.0) - 1.0);
}

static void w_i32(int32_t v) { fwrite(&v, 4, 1, g_out); }

static void write_tensor_header(const char* name, int n_dims, const int* ne, int ttype) {
    w_i32(n_dims); w_i32((int32_t)strlen(name)); w_i32(ttype);
    for (int i = 0; i < n_dims; ++i) w_i32(ne[i]);
    fwrite(name, 1, strlen(name), g_out);
}
/* write data given as f32 array; type 0 = f32, 1 = f16 */
static void write_tensor(const char* name, int n_dims, const int* ne, int ttype, const float* data) {
    write_tensor_header(name, n_dims, ne, ttype);
    size_t n = 1; for (int i = 0; i < n_dims; ++i) n *= (size_t)ne[i];
    if (ttype == 0) fwrite(data, 4, n, g_out);
    else { uint16_t* h = (uint16_t*)malloc(n * 2); for (size_t i = 0; i < n; ++i) h[i] = skw_f32_to_f16(data[i]); fwrite(h, 2, n, g_out); free(h); }
}
static float* gen_uniform(const char* name, size_t n, double scale) {
    float* d = (float*)malloc(n * 4); uint64_t key = g_seed * 0xD1342543DE82EF95ULL + fnv1a(name);
    for (size_t i = 0; i < n; ++i) d[i] = (float)(scale * urand(key, i));
    return d;
}
static int g_zero_in_from = -1; /* when >= 0: input columns >= this index are zeroed (layers never read the control dims) */
static int g_center_rows = 0; /* subtract each output row's mean: removes the static vector W*E[gelu] from MLP outputs */
static int g_zero_from = -1; /* when >= 0: output rows >= this index are zeroed (decoder control dims are never written by a layer) */
static void linear_w(const char* name, int out, int in, double gain) {
    int ne[2] = {in, out}; float* d = gen_uniform(name, (size_t)out * in, gain / sqrt((double)in));
    if (g_zero_from >= 0) for (int o = g_zero_from; o < out; ++o) for (int i = 0; i < in; ++i) d[(size_t)o * in + i] = 0.0f;
    if (g_center_rows) for (int o = 0; o < out; ++o) { double mu = 0; for (int i = 0; i < in; ++i) mu += d[(size_t)o * in + i]; mu /= in; for (int i = 0; i < in; ++i) d[(size_t)o * in + i] = (float)(d[(size_t)o * in + i] - mu); }
    if (g_zero_in_from >= 0) for (int o = 0; o < out; ++o) for (int i = g_zero_in_from; i < in; ++i) d[(size_t)o * in + i] = 0.0f;
    write_tensor(name, 2, ne, g_f16 ? 1 : 0, d); free(d);
}
static void vec_rand(const char* name, int n, double scale) { int ne[1] = {n}; float* d = gen_uniform(name, n, scale); if (g_zero_from >= 0) for (int o = g_zero_from; o < n; ++o) d[o] = 0.0f; write_tensor(name, 1, ne, 0, d); free(d); }
static void vec_ln_w(const char* name, int n) { int ne[1] = {n}; float* d = gen_uniform(name, n, 0.1); for (int i = 0; i < n; ++i) d[i] += 1.0f; write_tensor(name, 1, ne, 0, d); free(d); }

/* ---- Slaney mel filterbank (what whisper's mel_filters.npz holds) ---- */
static double hz_to_mel(double f) { const double f_sp = 200.0 / 3.0; if (f >= 1000.0) return 15.0 + log(f / 1000.0) / (log(6.4) / 27.0); return f / f_sp; }
static double mel_to_hz(double m) { const double f_sp = 200.0 / 3.0; if (m >= 15.0) return 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)); return f_sp * m; }
static void mel_filters(int n_mel, int n_fft_bins, float* out) {
    const double sr = 16000.0; const int n_fft = (n_fft_bins - 1) * 2;
    double* hz = (double*)malloc((n_mel + 2) * sizeof(double));
    double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mel + 2; ++i) hz[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mel + 1));
    for (int i = 0; i < n_mel; ++i) {
        double enorm = 2.0 / (hz[i + 2] - hz[i]);
        for (int k = 0; k < n_fft_bins; ++k) {
            double f = k * sr / n_fft;
            double lo = (f - hz[i]) / (hz[i + 1] - hz[i]), up = (hz[i + 2] - f) / (hz[i + 2] - hz[i + 1]);
            double w = lo < up ? lo : up; if (w < 0) w = 0;
            out[i * n_fft_bins + k] = (float)(w * enorm);
        }
    }
    free(hz);
}

/* ---- synthetic vocabulary ---- */
static const char* NST[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
    "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))",
    "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};
#define N_NST ((int)(sizeof(NST) / sizeof(NST[0])))
static void vocab_token(int i, char* buf) {
    if (i == 220) { strcpy(buf, " "); return; }
    if (i == 532) { strcpy(buf, " -"); return; }
    if (i == 705) { strcpy(buf, " '"); return; }
    if (i >= 1000 && i < 1000 + 2 * N_NST) { int j = i - 1000; if (j & 1) { buf[0] = ' '; strcpy(buf + 1, NST[j >> 1]); } else strcpy(buf, NST[j >> 1]); return; }
    if (i < 256) { /* single printable-ish bytes */ buf[0] = (char)(i < 94 ? 33 + i : 'a' + (i % 26)); buf[1] = (char)('a' + (i / 26) % 26); buf[2] = 0; if (i < 94) buf[1] = 0; return; }
    uint32_t h = (uint32_t)i * 2654435761u; int p = 0;
    if (h & 0x80000000u || (h & 3u)) buf[p++] = ' ';
    uint32_t v = (uint32_t)i; int len = 0; char tmp[16];
    do { tmp[len++] = (char)('a' + (v + (h >> (len * 3)) ) % 26); v /= 26; } while (v);
    /* keep it bijective: append the base-26 digits of i itself */
    v = (uint32_t)i; while (len < 3) tmp[len++] = 'a';
    for (int k = 0; k < len; ++k) buf[p++] = tmp[k];
    v = (uint32_t)i; do { buf[p++] = (char)('a' + v % 26); v /= 26; } while (v);
    buf[p] = 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s OUT.bin [--size tiny|base|small] [--vocab N] [--mels N] [--seed N] [--f32] [--knob value]...\n", argv[0]); return 2; }
    hparams_t hp = {51865, 1500, 768, 12, 12, 448, 768, 12, 12, 80, 1};
    for (int a = 2; a < argc; ++a) {
        if (!strcmp(argv[a], "--size") && a + 1 < argc) {
            const char* s = argv[++a];
            if (!strcmp(s, "tiny")) { hp.n_audio_state = hp.n_text_state = 384; hp.n_audio_head = hp.n_text_head = 6; hp.n_audio_layer = hp.n_text_layer = 4; }
            else if (!strcmp(s, "base")) { hp.n_audio_state = hp.n_text_state = 512; hp.n_audio_head = hp.n_text_head = 8; hp.n_audio_layer = hp.n_text_layer = 6; }
            else if (!strcmp(s, "small")) {}
            else if (!strcmp(s, "micro")) { hp.n_audio_state = hp.n_text_state = 128; hp.n_audio_head = hp.n_text_head = 2; hp.n_audio_layer = hp.n_text_layer = 2; }
            /* one-layer models of the other Whisper widths (base 512, medium 1024, large 1280): geometry coverage for kernels whose loop structure depends on d */
            else if (!strcmp(s, "w512")) { hp.n_audio_state = hp.n_text_state = 512; hp.n_audio_head = hp.n_text_head = 8; hp.n_audio_layer = hp.n_text_layer = 1; }
            else if (!strcmp(s, "w1024")) { hp.n_audio_state = hp.n_text_state = 1024; hp.n_audio_head = hp.n_text_head = 16; hp.n_audio_layer = hp.n_text_layer = 1; }
            else if (!strcmp(s, "w1280")) { hp.n_audio_state = hp.n_text_state = 1280; hp.n_audio_head = hp.n_text_head = 20; hp.n_audio_layer = hp.n_text_layer = 1; }
            else { fprintf(stderr, "unknown size %s\n", s); return 2; }
        } else if (!strcmp(argv[a], "--seed") && a + 1 < argc) g_seed = strtoull(argv[++a], NULL, 0);
        /* the other vocabularies and front ends of the model zoo: 51864 = the English-only files (*.en: the reference's default, plugins/native/whisper/src/lib.rs:68-70 — no language /
         * task tokens, every special id one lower), 51866 = large-v3 (one more language); --mels 128 = large-v3's filterbank */
        else if (!strcmp(argv[a], "--vocab") && a + 1 < argc) { hp.n_vocab = atoi(argv[++a]); if (hp.n_vocab < 51864 || hp.n_vocab > 51866) { fprintf(stderr, "--vocab 51864 | 51865 | 51866\n"); return 2; } }
        /* large-v3-turbo keeps large's 32 encoder layers and has 4 decoder layers: the two counts are independent header fields */
        else if (!strcmp(argv[a], "--audio-layers") && a + 1 < argc) hp.n_audio_layer = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--text-layers") && a + 1 < argc) hp.n_text_layer = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--mels") && a + 1 < argc) { hp.n_mels = atoi(argv[++a]); if (hp.n_mels != 80 && hp.n_mels != 128) { fprintf(stderr, "--mels 80 | 128\n"); return 2; } }
        else if (!strcmp(argv[a], "--f32")) g_f16 = 0;
        else if (!strncmp(argv[a], "--", 2) && a + 1 < argc) {
            int ok = 0; for (knob_t* k = knobs; k->name; ++k) if (!strcmp(k->name, argv[a] + 2)) { *k->v = atof(argv[++a]); ok = 1; break; }
            if (!ok) { fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
        } else { fprintf(stderr, "bad argument %s\n", argv[a]); return 2; }
    }
    hp.ftype = g_f16 ? 1 : 0;
    g_out = fopen(argv[1], "wb"); if (!g_out) { perror(argv[1]); return 1; }
    static char iobuf[1 << 22]; setvbuf(g_out, iobuf, _IOFBF, sizeof iobuf);

    /* magic + hparams */
    w_i32(0x67676d6c);
    w_i32(hp.n_vocab); w_i32(hp.n_audio_ctx); w_i32(hp.n_audio_state); w_i32(hp.n_audio_head); w_i32(hp.n_audio_layer);
    w_i32(hp.n_text_ctx); w_i32(hp.n_text_state); w_i32(hp.n_text_head); w_i32(hp.n_text_layer); w_i32(hp.n_mels); w_i32(hp.ftype);
    /* mel filters */
    { const int n_fft = 201; float* f = (float*)malloc(sizeof(float) * hp.n_mels * n_fft); mel_filters(hp.n_mels, n_fft, f); w_i32(hp.n_mels); w_i32(n_fft); fwrite(f, 4, (size_t)hp.n_mels * n_fft, g_out); free(f); }
    /* vocab: the BPE part only (multilingual: 50257 entries); specials are synthesised by the loader */
    { const int n = 50257; w_i32(n); char buf[64]; for (int i = 0; i < n; ++i) { vocab_token(i, buf); uint32_t len = (uint32_t)strlen(buf); fwrite(&len, 4, 1, g_out); fwrite(buf, 1, len, g_out); } }

    const int da = hp.n_audio_state, dt = hp.n_text_state; char nm[128];
    const int wtype = g_f16 ? 1 : 0;
    /* ---------------- encoder ---------------- */
    { /* sinusoidal positional embedding, as openai/whisper sinusoids() */
        int ne[2] = {da, hp.n_audio_ctx}; float* d = (float*)malloc(sizeof(float) * da * hp.n_audio_ctx);
        const int half = da / 2; const double inc = log(10000.0) / (half - 1);
        for (int t = 0; t < hp.n_audio_ctx; ++t) for (int i = 0; i < half; ++i) { double s = t * exp(-inc * i); d[t * da + i] = (float)sin(s); d[t * da + half + i] = (float)cos(s); }
        write_tensor("encoder.positional_embedding", 2, ne, 0, d); free(d);
    }
    { int ne[3] = {3, hp.n_mels, da}; float* d = gen_uniform("encoder.conv1.weight", (size_t)3 * hp.n_mels * da, k_conv1_gain / sqrt(3.0 * hp.n_mels)); write_tensor("encoder.conv1.weight", 3, ne, wtype, d); free(d); }
    { int ne[2] = {1, da}; float* d = gen_uniform("encoder.conv1.bias", da, 0.1); write_tensor("encoder.conv1.bias", 2, ne, 0, d); free(d); }
    { int ne[3] = {3, da, da}; float* d = gen_uniform("encoder.conv2.weight", (size_t)3 * da * da, k_conv2_gain / sqrt(3.0 * da)); write_tensor("encoder.conv2.weight", 3, ne, wtype, d); free(d); }
    { int ne[2] = {1, da}; float* d = gen_uniform("encoder.conv2.bias", da, 0.1); write_tensor("encoder.conv2.bias", 2, ne, 0, d); free(d); }
    vec_ln_w("encoder.ln_post.weight", da); vec_rand("encoder.ln_post.bias", da, 0.1);
    for (int l = 0; l < hp.n_audio_layer; ++l) {
#define ENC(fmt) (snprintf(nm, sizeof nm, "encoder.blocks.%d." fmt, l), nm)
        vec_ln_w(ENC("attn_ln.weight"), da); vec_rand(ENC("attn_ln.bias"), da, 0.1);
        linear_w(ENC("attn.query.weight"), da, da, 2.0 * k_enc_gain); vec_rand(ENC("attn.query.bias"), da, 0.1);
        linear_w(ENC("attn.key.weight"), da, da, 2.0 * k_enc_gain);
        linear_w(ENC("attn.value.weight"), da, da, k_enc_gain); vec_rand(ENC("attn.value.bias"), da, 0.1);
        linear_w(ENC("attn.out.weight"), da, da, k_enc_gain); vec_rand(ENC("attn.out.bias"), da, 0.1);
        vec_ln_w(ENC("mlp_ln.weight"), da); vec_rand(ENC("mlp_ln.bias"), da, 0.1);
        linear_w(ENC("mlp.0.weight"), 4 * da, da, k_enc_mlp_gain); vec_rand(ENC("mlp.0.bias"), 4 * da, 0.1);
        g_center_rows = 1; linear_w(ENC("mlp.2.weight"), da, 4 * da, k_enc_mlp_gain); g_center_rows = 0; vec_rand(ENC("mlp.2.bias"), da, 0.1);
    }
    /* ---------------- decoder ---------------- */
    const int R = 16, D0 = dt - R, c0 = D0, c1 = D0 + 1, ctc = D0 + 2, NF = 4;
    const double freqs[4] = {1.0, 3.0, 9.0, 27.0};
    const int multilingual = hp.n_vocab >= 51865;      /* whisper.cpp's is_multilingual(): the special ids sit one lower without it, and large-v3's extra language moves those after the languages up */
    const int tok_eot = multilingual ? 50257 : 50256, tok_beg = multilingual ? 50364 + (hp.n_vocab - 51865) : 50363;
    { /* positional embedding [n_text_ctx][dt] */
        int ne[2] = {dt, hp.n_text_ctx}; float* d = gen_uniform("decoder.positional_embedding", (size_t)dt * hp.n_text_ctx, k_pos_scale);
        for (int p = 0; p < hp.n_text_ctx; ++p) {
            float* row = d + (size_t)p * dt;
            for (int i = D0; i < dt; ++i) row[i] = 0.0f;
            double t = (p - k_eot_p0) / (k_eot_p1 - k_eot_p0); if (t < 0) t = 0; if (t > 1) t = 1;
            row[c0] = (float)(k_eot_lo + (k_eot_hi - k_eot_lo) * t);
            const int p_first = multilingual ? 2 : 0;      /* position of the last prompt token (sot, language, task | sot alone): the first sampled token's input */
            int q = p - p_first; int on = (q >= 0) && (fmod((double)q, k_ts_period) < 0.5);
            row[c1] = (float)(on ? k_ts_on : k_ts_off);
            double ktarget = k_ts_speed * (p - p_first); if (ktarget < 0) ktarget = 0;
            for (int m = 0; m < NF; ++m) { double w = 2.0 * M_PI * freqs[m] / 1501.0; row[ctc + 2 * m] = (float)(k_tc_amp * cos(w * ktarget)); row[ctc + 2 * m + 1] = (float)(k_tc_amp * sin(w * ktarget)); }
        }
        write_tensor("decoder.positional_embedding", 2, ne, 0, d); free(d);
    }
    { /* token embedding [n_vocab][dt] */
        int ne[2] = {dt, hp.n_vocab}; float* d = gen_uniform("decoder.token_embedding.weight", (size_t)dt * hp.n_vocab, k_emb_scale);
        for (int t = 0; t < hp.n_vocab; ++t) {
            float* row = d + (size_t)t * dt;
            for (int i = D0; i < dt; ++i) row[i] = 0.0f;
            if (t == tok_eot) row[c0] = (float)k_ctl_w;
            if (t >= tok_beg) {
                for (int i = 0; i < D0; ++i) row[i] = (float)(row[i] * k_ts_rho);
                row[c1] = (float)k_ctl_w; int k = t - tok_beg;
                for (int m = 0; m < NF; ++m) { double w = 2.0 * M_PI * freqs[m] / 1501.0; row[ctc + 2 * m] = (float)(k_ctl_w * cos(w * k)); row[ctc + 2 * m + 1] = (float)(k_ctl_w * sin(w * k)); }
            }
        }
        write_tensor("decoder.token_embedding.weight", 2, ne, wtype, d); free(d);
    }
    { int ne[1] = {dt}; float* d = (float*)malloc(4 * dt); for (int i = 0; i < dt; ++i) d[i] = (float)(i < D0 ? k_gamma_text : k_gamma_ctl); write_tensor("decoder.ln.weight", 1, ne, 0, d); for (int i = 0; i < dt; ++i) d[i] = 0.0f; write_tensor("decoder.ln.bias", 1, ne, 0, d); free(d); }
    for (int l = 0; l < hp.n_text_layer; ++l) {
#define DEC(fmt) (snprintf(nm, sizeof nm, "decoder.blocks.%d." fmt, l), nm)
        vec_ln_w(DEC("attn_ln.weight"), dt); vec_rand(DEC("attn_ln.bias"), dt, 0.1);
        g_zero_in_from = D0;
        linear_w(DEC("attn.query.weight"), dt, dt, k_dec_qk_gain); vec_rand(DEC("attn.query.bias"), dt, 0.1);
        linear_w(DEC("attn.key.weight"), dt, dt, k_dec_qk_gain);
        linear_w(DEC("attn.value.weight"), dt, dt, k_dec_vo_gain); vec_rand(DEC("attn.value.bias"), dt, 0.1);
        g_zero_in_from = -1;
        g_zero_from = D0; linear_w(DEC("attn.out.weight"), dt, dt, k_dec_vo_gain); vec_rand(DEC("attn.out.bias"), dt, 0.1); g_zero_from = -1;
        vec_ln_w(DEC("cross_attn_ln.weight"), dt); vec_rand(DEC("cross_attn_ln.bias"), dt, 0.1);
        g_zero_in_from = D0; linear_w(DEC("cross_attn.query.weight"), dt, dt, k_cross_qk_gain); g_zero_in_from = -1; vec_rand(DEC("cross_attn.query.bias"), dt, 0.1);
        linear_w(DEC("cross_attn.key.weight"), dt, da, k_cross_qk_gain);
        linear_w(DEC("cross_attn.value.weight"), dt, da, k_cross_vo_gain); vec_rand(DEC("cross_attn.value.bias"), dt, 0.1);
        g_zero_from = D0; linear_w(DEC("cross_attn.out.weight"), dt, dt, k_cross_vo_gain); vec_rand(DEC("cross_attn.out.bias"), dt, 0.1); g_zero_from = -1;
        vec_ln_w(DEC("mlp_ln.weight"), dt); vec_rand(DEC("mlp_ln.bias"), dt, 0.1);
        g_zero_in_from = D0; linear_w(DEC("mlp.0.weight"), 4 * dt, dt, k_dec_fc1_gain); g_zero_in_from = -1; vec_rand(DEC("mlp.0.bias"), 4 * dt, 0.1);
        g_zero_from = D0; g_center_rows = 1; linear_w(DEC("mlp.2.weight"), dt, 4 * dt, k_dec_fc2_gain); g_center_rows = 0; vec_rand(DEC("mlp.2.bias"), dt, 0.1); g_zero_from = -1;
    }
    if (fclose(g_out) != 0) { perror("fclose"); return 1; }
    return 0;
}
