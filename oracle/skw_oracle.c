/*
 * oracle/skw_oracle.c — TEST INFRASTRUCTURE (see skw_oracle.h header: PARITY UNPINNED).
 *
 * CPU restatement of whisper.cpp's algorithm as the reference node invokes it
 * (/root/reference/plugins/native/whisper/src/lib.rs:624-646: Greedy{best_of:1},
 * language, translate=false, suppress_blank, suppress_nst, then `full`).
 * Each function cites the whisper.cpp routine it restates; numerics follow
 * include/skw_math.h (f16-rounded operands, k-ascending fma chains, double
 * accumulation where ggml uses ggml_float).
 *
 * Build: see oracle/Makefile (-O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#define _GNU_SOURCE
#include "skw_oracle.h"
#include "../include/skw_math.h"
#include "../include/skw_ggml_quant.h"
#include <alloca.h>
#include <immintrin.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WHISPER_SAMPLE_RATE 16000
#define WHISPER_N_FFT 400
#define WHISPER_HOP_LENGTH 160
#define WHISPER_CHUNK_SIZE 30
#define DELTA_MIN 10   /* whisper_full_with_state: `const int delta_min = 10` mel frames = 100 ms (whisper.cpp #2065; the pre-1.6 rule was 100 = 1 s) */
#define N_LANG 99

/* ------------------------------------------------------------------ model */
typedef struct { float* wt; float* b; int n_in, n_out;
                 int qtype; int8_t* qw; float* qd; float* qm; /* ggml q8 arithmetic (include/skw_ggml_quant.h (a)): qw [n_out][n_in], qd / qm [n_out][n_in / 32]; NULL for f16 weights */
} lin_t; /* wt: [n_in][n_out] (transposed), b may be NULL */
typedef struct { float *w, *b; } ln_t;
typedef struct { ln_t attn_ln, mlp_ln; lin_t q, k, v, o, fc1, fc2; } enc_layer_t;
typedef struct { ln_t attn_ln, cross_ln, mlp_ln; lin_t q, k, v, o, cq, ck, cv, co, fc1, fc2; } dec_layer_t;

struct skwo_model {
    skwo_hparams hp;
    int n_mel_f, n_fft_f; float* filters;
    int n_vocab_file; char** tok_str; int* tok_len;
    int tok_eot, tok_sot, tok_translate, tok_transcribe, tok_solm, tok_prev, tok_nosp, tok_not, tok_beg;
    int tok_space, tok_sp_dash, tok_sp_quote; int* nst_ids; int n_nst;
    /* encoder */
    float* e_pe; lin_t conv1, conv2; ln_t ln_post; enc_layer_t* enc;
    /* decoder */
    float* d_pe; float* d_te; /* [n_vocab][d] natural layout for embedding lookup */ lin_t d_te_lin; ln_t d_ln; dec_layer_t* dec;
    int quant;   /* ggml type of the matmul weights when the file is uniformly block-quantised and ggml's q8 arithmetic is on; 0 = f16 arithmetic */
    uint16_t* gelu_tab; float sin_vals[WHISPER_N_FFT], cos_vals[WHISPER_N_FFT], hann[WHISPER_N_FFT];
};

/* raw tensor as read from file */
typedef struct { char name[96]; int n_dims; int ne[4]; int type; size_t n; float* data; uint8_t* qblk; int qtype; } raw_t;   /* qblk: the file's blocks of a quantised tensor */

static float* xmalloc_f(size_t n) { float* p = (float*)aligned_alloc(64, ((n * 4 + 63) / 64) * 64); if (!p) { fprintf(stderr, "oracle: oom\n"); abort(); } return p; }

static const char* NST_LIST[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
    "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))",
    "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac",
        "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};
#define N_NST_LIST ((int)(sizeof(NST_LIST) / sizeof(NST_LIST[0])))

/* 1 (default): a uniformly quantised file runs ggml's q8 arithmetic; 0: its dequantised f16 twin (what the f16_mfma precision of the engine runs) */
static int g_quant_mode = 1;
void skwo_set_quant_mode(int mode) { g_quant_mode = mode; }
static int is_matmul_weight(const raw_t* t) {
    const size_t L = strlen(t->name);
    return t->n_dims == 2 && L > 7 && !strcmp(t->name + L - 7, ".weight") && !strstr(t->name, "positional_embedding") && !strstr(t->name, "ln");
}
static raw_t* find_t(raw_t* ts, int n, const char* name) { for (int i = 0; i < n; ++i) if (!strcmp(ts[i].name, name)) return &ts[i]; return NULL; }

static int take_lin(raw_t* ts, int nt, const char* wname, const char* bname, lin_t* L, char* err, int errlen) {
    raw_t* w = find_t(ts, nt, wname);
    if (!w) { snprintf(err, errlen, "missing tensor %s", wname); return -1; }
    if (w->type != 1) { snprintf(err, errlen, "tensor %s: only f16 matmul weights are supported (type %d)", wname, w->type); return -1; }
    int n_in, n_out;
    if (w->n_dims == 2) { n_in = w->ne[0]; n_out = w->ne[1]; }
    else if (w->n_dims == 3) { n_in = w->ne[0] * w->ne[1]; n_out = w->ne[2]; }
    else { snprintf(err, errlen, "tensor %s: bad dims", wname); return -1; }
    L->n_in = n_in; L->n_out = n_out; L->wt = xmalloc_f((size_t)n_in * n_out);
    if (w->n_dims == 2) {
        for (int o = 0; o < n_out; ++o) for (int i = 0; i < n_in; ++i) L->wt[(size_t)i * n_out + o] = w->data[(size_t)o * n_in + i];
    } else { /* conv weight [oc][ic][kw] -> k = tap*IC + ic */
        int kw = w->ne[0], ic = w->ne[1];
        for (int o = 0; o < n_out; ++o) for (int c = 0; c < ic; ++c) for (int t = 0; t < kw; ++t)
            L->wt[(size_t)(t * ic + c) * n_out + o] = w->data[((size_t)o * ic + c) * kw + t];
    }
    L->qtype = 0; L->qw = NULL; L->qd = L->qm = NULL;
    if (w->qblk && w->qtype) {   /* ggml q8 arithmetic: keep the blocks in the common integer form */
        const int nb = n_in / 32; const size_t bb = skw_ggml_block_bytes(w->qtype);
        L->qtype = w->qtype; L->qw = (int8_t*)malloc((size_t)n_out * n_in); L->qd = xmalloc_f((size_t)n_out * nb); L->qm = xmalloc_f((size_t)n_out * nb);
        for (size_t i = 0; i < (size_t)n_out * nb; ++i) skw_ggml_unpack_block(w->qtype, w->qblk + i * bb, L->qw + i * 32, &L->qd[i], &L->qm[i]);
    }
    L->b = NULL;
    if (bname) { raw_t* b = find_t(ts, nt, bname); if (!b) { snprintf(err, errlen, "missing tensor %s", bname); return -1; } L->b = b->data; b->data = NULL; }
    return 0;
}
static int take_ln(raw_t* ts, int nt, const char* wname, const char* bname, ln_t* L, char* err, int errlen) {
    raw_t* w = find_t(ts, nt, wname); raw_t* b = find_t(ts, nt, bname);
    if (!w || !b) { snprintf(err, errlen, "missing tensor %s/%s", wname, bname); return -1; }
    L->w = w->data; w->data = NULL; L->b = b->data; b->data = NULL; return 0;
}

skwo_model* skwo_load(const char* path, char* err, int errlen) {
    FILE* f = fopen(path, "rb");
    if (!f) { snprintf(err, errlen, "cannot open %s", path); return NULL; }
    int32_t magic; if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6c) { snprintf(err, errlen, "bad magic"); fclose(f); return NULL; }
    skwo_model* m = (skwo_model*)calloc(1, sizeof *m);
    if (fread(&m->hp, 4, 11, f) != 11) { snprintf(err, errlen, "short hparams"); fclose(f); free(m); return NULL; }
    int32_t nm, nf; if (fread(&nm, 4, 1, f) != 1 || fread(&nf, 4, 1, f) != 1) { snprintf(err, errlen, "short filters"); fclose(f); free(m); return NULL; }
    m->n_mel_f = nm; m->n_fft_f = nf; m->filters = xmalloc_f((size_t)nm * nf);
    if (fread(m->filters, 4, (size_t)nm * nf, f) != (size_t)nm * nf) { snprintf(err, errlen, "short filters"); fclose(f); free(m); return NULL; }
    int32_t nv; if (fread(&nv, 4, 1, f) != 1) { snprintf(err, errlen, "short vocab"); fclose(f); free(m); return NULL; }
    m->n_vocab_file = nv; int NV = m->hp.n_vocab;
    m->tok_str = (char**)calloc(NV, sizeof(char*)); m->tok_len = (int*)calloc(NV, sizeof(int));
    for (int i = 0; i < nv; ++i) {
        uint32_t len; if (fread(&len, 4, 1, f) != 1) { snprintf(err, errlen, "short vocab"); fclose(f); return NULL; }
        char* s = (char*)malloc(len + 1); if (len && fread(s, 1, len, f) != len) { snprintf(err, errlen, "short vocab"); fclose(f); return NULL; }
        s[len] = 0; if (i < NV) { m->tok_str[i] = s; m->tok_len[i] = (int)len; } else free(s);
    }
    /* special token ids (whisper.cpp whisper_vocab + is_multilingual shift) */
    m->tok_eot = 50256; m->tok_sot = 50257; m->tok_translate = 50357; m->tok_transcribe = 50358; m->tok_solm = 50359;
    m->tok_prev = 50360; m->tok_nosp = 50361; m->tok_not = 50362; m->tok_beg = 50363;
    if (NV >= 51865) {
        m->tok_eot++; m->tok_sot++;
        int dt = NV - 51865; /* large-v3 adds one language */
        m->tok_translate += 1 + dt; m->tok_transcribe += 1 + dt; m->tok_solm += 1 + dt; m->tok_prev += 1 + dt; m->tok_nosp += 1 + dt; m->tok_not += 1 + dt; m->tok_beg += 1 + dt;
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
        else if (i > m->tok_sot && i <= m->tok_sot + (NV - 51865 + N_LANG)) snprintf(buf, sizeof buf, "[_LANG_%d]", i - m->tok_sot - 1);
        else snprintf(buf, sizeof buf, "[_extra_token_%d]", i);
        m->tok_str[i] = strdup(buf); m->tok_len[i] = (int)strlen(buf);
    }
    /* token_to_id lookups that whisper_process_logits performs by string */
    m->tok_space = m->tok_sp_dash = m->tok_sp_quote = -1;
    m->nst_ids = (int*)malloc(sizeof(int) * 2 * N_NST_LIST); m->n_nst = 0;
    for (int i = 0; i < NV; ++i) {
        const char* s = m->tok_str[i]; if (!s) continue;
        if (!strcmp(s, " ")) m->tok_space = i; /* later duplicates overwrite, as std::map::operator[] would */
        if (!strcmp(s, " -")) m->tok_sp_dash = i;
        if (!strcmp(s, " '")) m->tok_sp_quote = i;
    }
    for (int j = 0; j < N_NST_LIST; ++j) for (int sp = 0; sp < 2; ++sp) {
        char buf[32]; snprintf(buf, sizeof buf, "%s%s", sp ? " " : "", NST_LIST[j]);
        int id = -1; for (int i = 0; i < NV; ++i) if (m->tok_str[i] && !strcmp(m->tok_str[i], buf)) id = i;
        if (id >= 0) m->nst_ids[m->n_nst++] = id;
    }
    /* tensors */
    int cap = 512, nt = 0; raw_t* ts = (raw_t*)calloc(cap, sizeof(raw_t));
    for (;;) {
        int32_t nd, len, tt; if (fread(&nd, 4, 1, f) != 1) break;
        if (fread(&len, 4, 1, f) != 1 || fread(&tt, 4, 1, f) != 1) break;
        if (nt == cap) { cap *= 2; ts = (raw_t*)realloc(ts, cap * sizeof(raw_t)); memset(ts + nt, 0, (cap - nt) * sizeof(raw_t)); }
        raw_t* t = &ts[nt]; t->n_dims = nd; t->type = tt; t->n = 1;
        for (int i = 0; i < 4; ++i) t->ne[i] = 1;
        for (int i = 0; i < nd; ++i) { int32_t e; if (fread(&e, 4, 1, f) != 1) { snprintf(err, errlen, "short tensor header"); return NULL; } t->ne[i] = e; t->n *= (size_t)e; }
        if (len >= (int)sizeof t->name) { snprintf(err, errlen, "tensor name too long"); return NULL; }
        if (fread(t->name, 1, len, f) != (size_t)len) { snprintf(err, errlen, "short tensor name"); return NULL; }
        t->name[len] = 0; t->data = xmalloc_f(t->n);
        if (tt == 0) { if (fread(t->data, 4, t->n, f) != t->n) { snprintf(err, errlen, "short tensor data %s", t->name); return NULL; } }
        else if (tt == 1) { uint16_t* h = (uint16_t*)malloc(t->n * 2);
        if (fread(h, 2, t->n, f) != t->n) { snprintf(err, errlen, "short tensor data %s", t->name);
        return NULL; } for (size_t i = 0; i < t->n; ++i) t->data[i] = skw_f16_to_f32(h[i]); free(h); }
        else if (skw_ggml_block_bytes(tt) && t->ne[0] % 32 == 0) {   /* block-quantised: blocks kept, decoded once the whole file has been seen */
            const size_t bb = skw_ggml_block_bytes(tt), nb = t->n / 32; t->qblk = (uint8_t*)malloc(nb * bb); t->qtype = tt;
            if (fread(t->qblk, bb, nb, f) != nb) { snprintf(err, errlen, "short tensor data %s", t->name); return NULL; }
        }
        else { snprintf(err, errlen, "tensor %s: unsupported ggml type %d (f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0 are read)", t->name, tt); return NULL; }
        nt++;
    }
    fclose(f);
    /* ggml's q8 arithmetic needs every matmul weight in one block type (what whisper.cpp's quantize writes); anything else runs as the f16 twin */
    { int qt = 0, uniform = g_quant_mode != 0;
      for (int i = 0; i < nt; ++i) if (is_matmul_weight(&ts[i])) { if (!ts[i].qblk) uniform = 0; else if (!qt) qt = ts[i].qtype; else if (qt != ts[i].qtype) uniform = 0; }
      m->quant = (uniform && qt) ? qt : 0;
      for (int i = 0; i < nt; ++i) if (ts[i].qblk) {
          raw_t* t = &ts[i]; const size_t bb = skw_ggml_block_bytes(t->qtype), nb = t->n / 32;
          for (size_t b = 0; b < nb; ++b) { skw_ggml_dequant_block(t->qtype, t->qblk + b * bb, t->data + b * 32);
          if (!m->quant) for (int j = 0; j < 32; ++j) t->data[b * 32 + j] = skw_round_f16(t->data[b * 32 + j]); }
          t->type = 1; if (!m->quant || !is_matmul_weight(t)) { free(t->qblk); t->qblk = NULL; t->qtype = 0; }
      } }
    int rc = 0; char nmw[128], nmb[128];
    raw_t* t;
    if (!(t = find_t(ts, nt, "encoder.positional_embedding"))) { snprintf(err, errlen, "missing encoder.positional_embedding"); return NULL; } m->e_pe = t->data; t->data = NULL;
    if (!(t = find_t(ts, nt, "decoder.positional_embedding"))) { snprintf(err, errlen, "missing decoder.positional_embedding"); return NULL; } m->d_pe = t->data; t->data = NULL;
    rc |= take_lin(ts, nt, "encoder.conv1.weight", "encoder.conv1.bias", &m->conv1, err, errlen);
    rc |= take_lin(ts, nt, "encoder.conv2.weight", "encoder.conv2.bias", &m->conv2, err, errlen);
    rc |= take_ln(ts, nt, "encoder.ln_post.weight", "encoder.ln_post.bias", &m->ln_post, err, errlen);
    if (rc) return NULL;
    m->enc = (enc_layer_t*)calloc(m->hp.n_audio_layer, sizeof(enc_layer_t));
    for (int l = 0; l < m->hp.n_audio_layer && !rc; ++l) {
        enc_layer_t* L = &m->enc[l];
#define NW(fmt) (snprintf(nmw, sizeof nmw, "encoder.blocks.%d." fmt ".weight", l), nmw)
#define NB(fmt) (snprintf(nmb, sizeof nmb, "encoder.blocks.%d." fmt ".bias", l), nmb)
        rc |= take_ln(ts, nt, NW("attn_ln"), NB("attn_ln"), &L->attn_ln, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.query"), NB("attn.query"), &L->q, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.key"), NULL, &L->k, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.value"), NB("attn.value"), &L->v, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.out"), NB("attn.out"), &L->o, err, errlen);
        rc |= take_ln(ts, nt, NW("mlp_ln"), NB("mlp_ln"), &L->mlp_ln, err, errlen);
        rc |= take_lin(ts, nt, NW("mlp.0"), NB("mlp.0"), &L->fc1, err, errlen);
        rc |= take_lin(ts, nt, NW("mlp.2"), NB("mlp.2"), &L->fc2, err, errlen);
#undef NW
#undef NB
    }
    if (rc) return NULL;
    if (!(t = find_t(ts, nt, "decoder.token_embedding.weight")) || t->type != 1) { snprintf(err, errlen, "missing/unsupported decoder.token_embedding.weight"); return NULL; }
    { int d = m->hp.n_text_state; m->d_te = xmalloc_f(t->n); memcpy(m->d_te, t->data, t->n * 4); rc |= take_lin(ts, nt, "decoder.token_embedding.weight", NULL, &m->d_te_lin, err, errlen); (void)d; }
    rc |= take_ln(ts, nt, "decoder.ln.weight", "decoder.ln.bias", &m->d_ln, err, errlen);
    m->dec = (dec_layer_t*)calloc(m->hp.n_text_layer, sizeof(dec_layer_t));
    for (int l = 0; l < m->hp.n_text_layer && !rc; ++l) {
        dec_layer_t* L = &m->dec[l];
#define NW(fmt) (snprintf(nmw, sizeof nmw, "decoder.blocks.%d." fmt ".weight", l), nmw)
#define NB(fmt) (snprintf(nmb, sizeof nmb, "decoder.blocks.%d." fmt ".bias", l), nmb)
        rc |= take_ln(ts, nt, NW("attn_ln"), NB("attn_ln"), &L->attn_ln, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.query"), NB("attn.query"), &L->q, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.key"), NULL, &L->k, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.value"), NB("attn.value"), &L->v, err, errlen);
        rc |= take_lin(ts, nt, NW("attn.out"), NB("attn.out"), &L->o, err, errlen);
        rc |= take_ln(ts, nt, NW("cross_attn_ln"), NB("cross_attn_ln"), &L->cross_ln, err, errlen);
        rc |= take_lin(ts, nt, NW("cross_attn.query"), NB("cross_attn.query"), &L->cq, err, errlen);
        rc |= take_lin(ts, nt, NW("cross_attn.key"), NULL, &L->ck, err, errlen);
        rc |= take_lin(ts, nt, NW("cross_attn.value"), NB("cross_attn.value"), &L->cv, err, errlen);
        rc |= take_lin(ts, nt, NW("cross_attn.out"), NB("cross_attn.out"), &L->co, err, errlen);
        rc |= take_ln(ts, nt, NW("mlp_ln"), NB("mlp_ln"), &L->mlp_ln, err, errlen);
        rc |= take_lin(ts, nt, NW("mlp.0"), NB("mlp.0"), &L->fc1, err, errlen);
        rc |= take_lin(ts, nt, NW("mlp.2"), NB("mlp.2"), &L->fc2, err, errlen);
#undef NW
#undef NB
    }
    if (rc) return NULL;
    for (int i = 0; i < nt; ++i) { free(ts[i].data); free(ts[i].qblk); }
    free(ts);
    /* ggml_table_gelu_f16 */
    m->gelu_tab = (uint16_t*)malloc(65536 * 2); for (int i = 0; i < 65536; ++i) m->gelu_tab[i] = skw_gelu_table_entry((uint16_t)i);
    /* whisper_global_cache: sin/cos table and periodic Hann window */
    for (int i = 0; i < WHISPER_N_FFT; ++i) {
        double theta = (2 * M_PI * i) / WHISPER_N_FFT;
        m->sin_vals[i] = sinf(theta); m->cos_vals[i] = cosf(theta);
        m->hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / (WHISPER_N_FFT)));
    }
    return m;
}

static void free_lin(lin_t* L) { free(L->wt); free(L->b); free(L->qw); free(L->qd); free(L->qm); }
static void free_ln(ln_t* L) { free(L->w); free(L->b); }
void skwo_free(skwo_model* m) {
    if (!m) return;
    for (int i = 0; i < m->hp.n_vocab; ++i) free(m->tok_str[i]);
    free(m->tok_str); free(m->tok_len); free(m->nst_ids); free(m->filters); free(m->e_pe); free(m->d_pe); free(m->d_te); free(m->gelu_tab);
    free_lin(&m->conv1); free_lin(&m->conv2); free_ln(&m->ln_post); free_lin(&m->d_te_lin); free_ln(&m->d_ln);
    for (int l = 0; l < m->hp.n_audio_layer; ++l) { enc_layer_t* L = &m->enc[l]; free_ln(&L->attn_ln); free_ln(&L->mlp_ln); free_lin(&L->q); free_lin(&L->k);
    free_lin(&L->v); free_lin(&L->o); free_lin(&L->fc1); free_lin(&L->fc2); }
    for (int l = 0; l < m->hp.n_text_layer; ++l) { dec_layer_t* L = &m->dec[l]; free_ln(&L->attn_ln); free_ln(&L->cross_ln); free_ln(&L->mlp_ln);
    free_lin(&L->q); free_lin(&L->k); free_lin(&L->v); free_lin(&L->o); free_lin(&L->cq); free_lin(&L->ck); free_lin(&L->cv); free_lin(&L->co);
    free_lin(&L->fc1); free_lin(&L->fc2); }
    free(m->enc); free(m->dec); free(m);
}
void skwo_get_hparams(const skwo_model* m, skwo_hparams* out) { *out = m->hp; }
int skwo_model_quant(const skwo_model* m) { return m->quant; }
const char* skwo_token_str(const skwo_model* m, int id, int* len) { if (id < 0 || id >= m->hp.n_vocab) { if (len) *len = 0; return ""; } if (len) *len = m->tok_len[id]; return m->tok_str[id]; }
void skwo_free_buf(void* p) { free(p); }
void skwo_default_params(skwo_params* p) {
    memset(p, 0, sizeof *p);
    p->lang_id = 0; p->translate = 0; p->suppress_blank = 1; p->suppress_nst = 0; /* whisper_full_default_params */ p->no_timestamps = 0; p->single_segment = 0; p->max_tokens = 0;
    p->max_initial_ts = 1.0f; p->entropy_thold = 2.4f; p->logprob_thold = -1.0f; p->no_speech_thold = 0.6f; p->n_threads = 0;
    p->temperature = 0.0f; p->temperature_inc = 0.2f;
}

/* --------------------------------------------------------- K1: log-mel */
/* whisper.cpp dft(): naive DFT on table lookups, f32, separate mul and add roundings */
static void o_dft(const skwo_model* m, const float* in, int N, float* out) {
    const int step = WHISPER_N_FFT / N;
    for (int k = 0; k < N; k++) {
        float re = 0, im = 0;
        for (int n = 0; n < N; n++) {
            int idx = (k * n * step) % WHISPER_N_FFT;
            re += in[n] * m->cos_vals[idx];
            im -= in[n] * m->sin_vals[idx];
        }
        out[k * 2 + 0] = re; out[k * 2 + 1] = im;
    }
}
/* whisper.cpp fft(): recursive radix-2, falling back to dft() for odd N; `in` has room for the recursion */
static void o_fft(const skwo_model* m, float* in, int N, float* out) {
    if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
    const int half_N = N / 2;
    if (N - half_N * 2 == 1) { o_dft(m, in, N, out); return; }
    float* even = in + N;
    for (int i = 0; i < half_N; ++i) even[i] = in[2 * i];
    float* even_fft = out + 2 * N;
    o_fft(m, even, half_N, even_fft);
    float* odd = even;
    for (int i = 0; i < half_N; ++i) odd[i] = in[2 * i + 1];
    float* odd_fft = even_fft + N;
    o_fft(m, odd, half_N, odd_fft);
    const int step = WHISPER_N_FFT / N;
    for (int k = 0; k < half_N; k++) {
        int idx = k * step;
        float re = m->cos_vals[idx], im = -m->sin_vals[idx];
        float re_odd = odd_fft[2 * k + 0], im_odd = odd_fft[2 * k + 1];
        out[2 * k + 0] = even_fft[2 * k + 0] + re * re_odd - im * im_odd;
        out[2 * k + 1] = even_fft[2 * k + 1] + re * im_odd + im * re_odd;
        out[2 * (k + half_N) + 0] = even_fft[2 * k + 0] - re * re_odd + im * im_odd;
        out[2 * (k + half_N) + 1] = even_fft[2 * k + 1] - re * im_odd - im * re_odd;
    }
}

/* whisper.cpp log_mel_spectrogram + log_mel_spectrogram_worker_thread */
float* skwo_log_mel(const skwo_model* m, const float* pcm, int n_samples, int* n_len_out, int* n_len_org_out) {
    const int frame_size = WHISPER_N_FFT, frame_step = WHISPER_HOP_LENGTH, n_mel = m->hp.n_mels, n_fft = m->n_fft_f;
    const int64_t stage_1_pad = WHISPER_SAMPLE_RATE * 30, stage_2_pad = frame_size / 2;
    const int64_t padded_n = n_samples + stage_1_pad + stage_2_pad * 2;
    float* sp = (float*)calloc(padded_n, sizeof(float));
    memcpy(sp + stage_2_pad, pcm, sizeof(float) * n_samples);
    /* reflective pad at the beginning: reverse_copy(samples+1, samples+1+200, padded.begin()) */
    for (int i = 0; i < stage_2_pad; ++i) sp[i] = pcm[stage_2_pad - i];
    const int n_len = (int)((padded_n - frame_size) / frame_step);
    const int n_len_org = 1 + (int)((n_samples + stage_2_pad - frame_size) / frame_step);
    float* mel = (float*)malloc(sizeof(float) * (size_t)n_mel * n_len);
    const int ns = n_samples + (int)stage_2_pad; /* worker's n_samples */
    int n_calc = ns / frame_step + 1; if (n_calc > n_len) n_calc = n_len;
#pragma omp parallel
    {
        float* fft_in = (float*)calloc(frame_size * 2, sizeof(float));
        float* fft_out = (float*)calloc(frame_size * 2 * 2 * 2, sizeof(float));
#pragma omp for schedule(static)
        for (int i = 0; i < n_calc; ++i) {
            const int offset = i * frame_step;
            int lim = ns - offset; if (lim > frame_size) lim = frame_size;
            for (int j = 0; j < lim; ++j) fft_in[j] = m->hann[j] * sp[offset + j];
            for (int j = lim < 0 ? 0 : lim; j < frame_size; ++j) fft_in[j] = 0.0f;
            o_fft(m, fft_in, frame_size, fft_out);
            for (int j = 0; j < n_fft; ++j) fft_out[j] = (fft_out[2 * j + 0] * fft_out[2 * j + 0] + fft_out[2 * j + 1] * fft_out[2 * j + 1]);
            for (int j = 0; j < n_mel; ++j) {
                double sum = 0.0; int k = 0; const float* fl = m->filters + (size_t)j * n_fft;
                for (k = 0; k < n_fft - 3; k += 4)
                    sum += fft_out[k + 0] * fl[k + 0] + fft_out[k + 1] * fl[k + 1] + fft_out[k + 2] * fl[k + 2] + fft_out[k + 3] * fl[k + 3];
                for (; k < n_fft; k++) sum += fft_out[k] * fl[k];
                sum = log10(sum > 1e-10 ? sum : 1e-10);
                mel[(size_t)j * n_len + i] = (float)sum;
            }
        }
        free(fft_in); free(fft_out);
    }
    { double v = log10(1e-10); for (int i = n_calc; i < n_len; ++i) for (int j = 0; j < n_mel; ++j) mel[(size_t)j * n_len + i] = (float)v; }
    /* clamping and normalization */
    double mmax = -1e20;
    for (size_t i = 0; i < (size_t)n_mel * n_len; ++i) if (mel[i] > mmax) mmax = mel[i];
    mmax -= 8.0;
    for (size_t i = 0; i < (size_t)n_mel * n_len; ++i) { if (mel[i] < mmax) mel[i] = (float)mmax; mel[i] = (float)((mel[i] + 4.0) / 4.0); }
    free(sp);
    *n_len_out = n_len; *n_len_org_out = n_len_org;
    return mel;
}

/* ------------------------------------------------------- debug taps */
#define MAX_TAPS 64
static struct { char name[32]; float* data; size_t n; } g_taps[MAX_TAPS];
static int g_n_taps = 0, g_taps_on = 0;
void skwo_debug_enable(int on) { g_taps_on = on; for (int i = 0; i < g_n_taps; ++i) free(g_taps[i].data); g_n_taps = 0; }
static void tap(const char* name, const float* d, size_t n) {
    if (!g_taps_on || g_n_taps == MAX_TAPS) return;
    snprintf(g_taps[g_n_taps].name, 32, "%s", name); g_taps[g_n_taps].data = (float*)malloc(n * 4); memcpy(g_taps[g_n_taps].data, d, n * 4); g_taps[g_n_taps].n = n; g_n_taps++;
}
long skwo_debug_get(const char* name, float* out, size_t cap) {
    for (int i = 0; i < g_n_taps; ++i) if (!strcmp(g_taps[i].name, name)) { if (out && cap >= g_taps[i].n) memcpy(out, g_taps[i].data, g_taps[i].n * 4); return (long)g_taps[i].n; }
    return -1;
}

/* ------------------------------------------------------- arithmetic core */
/* C[m][n] = k-ascending chain acc = fma(A[m][k], Wt[k][n], acc), acc0 = 0.
 * A must already hold f16-representable values (ggml converts src1 rows to f16 before ggml_vec_dot_f16). */
static void gemm_chain(const float* A, long lda, int M, const float* Wt, long ldw, int N, int K, float* C, long ldc) {
    const int NB = 24;
    const int n_full = (N / 8) * 8;
    const int n_tiles = (n_full + NB - 1) / NB;
    const int m_tiles = (M + 3) / 4;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int mt = 0; mt < m_tiles; ++mt) for (int ntile = 0; ntile < n_tiles; ++ntile) {
        const int m0 = mt * 4, mr = (M - m0) < 4 ? (M - m0) : 4;
        const int n0 = ntile * NB; int nv = (n_full - n0) / 8; if (nv > 3) nv = 3;
        __m256 acc[4][3];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) acc[i][j] = _mm256_setzero_ps();
        const float* a0 = A + (long)m0 * lda; const float* a1 = A + (long)(m0 + (mr > 1 ? 1 : 0)) * lda;
        const float* a2 = A + (long)(m0 + (mr > 2 ? 2 : 0)) * lda; const float* a3 = A + (long)(m0 + (mr > 3 ? 3 : 0)) * lda;
        if (nv == 3) {
            for (int k = 0; k < K; ++k) {
                const float* w = Wt + (long)k * ldw + n0;
                __m256 w0 = _mm256_loadu_ps(w), w1 = _mm256_loadu_ps(w + 8), w2 = _mm256_loadu_ps(w + 16);
                __m256 x;
                x = _mm256_broadcast_ss(a0 + k); acc[0][0] = _mm256_fmadd_ps(x, w0, acc[0][0]); acc[0][1] = _mm256_fmadd_ps(x, w1, acc[0][1]); acc[0][2] = _mm256_fmadd_ps(x, w2, acc[0][2]);
                x = _mm256_broadcast_ss(a1 + k); acc[1][0] = _mm256_fmadd_ps(x, w0, acc[1][0]); acc[1][1] = _mm256_fmadd_ps(x, w1, acc[1][1]); acc[1][2] = _mm256_fmadd_ps(x, w2, acc[1][2]);
                x = _mm256_broadcast_ss(a2 + k); acc[2][0] = _mm256_fmadd_ps(x, w0, acc[2][0]); acc[2][1] = _mm256_fmadd_ps(x, w1, acc[2][1]); acc[2][2] = _mm256_fmadd_ps(x, w2, acc[2][2]);
                x = _mm256_broadcast_ss(a3 + k); acc[3][0] = _mm256_fmadd_ps(x, w0, acc[3][0]); acc[3][1] = _mm256_fmadd_ps(x, w1, acc[3][1]); acc[3][2] = _mm256_fmadd_ps(x, w2, acc[3][2]);
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const float* w = Wt + (long)k * ldw + n0;
                for (int j = 0; j < nv; ++j) {
                    __m256 wj = _mm256_loadu_ps(w + 8 * j);
                    acc[0][j] = _mm256_fmadd_ps(_mm256_broadcast_ss(a0 + k), wj, acc[0][j]);
                    acc[1][j] = _mm256_fmadd_ps(_mm256_broadcast_ss(a1 + k), wj, acc[1][j]);
                    acc[2][j] = _mm256_fmadd_ps(_mm256_broadcast_ss(a2 + k), wj, acc[2][j]);
                    acc[3][j] = _mm256_fmadd_ps(_mm256_broadcast_ss(a3 + k), wj, acc[3][j]);
                }
            }
        }
        for (int i = 0; i < mr; ++i) for (int j = 0; j < nv; ++j) _mm256_storeu_ps(C + (long)(m0 + i) * ldc + n0 + 8 * j, acc[i][j]);
    }
    if (n_full < N) { /* scalar tail columns: the same chain, one lane */
#pragma omp parallel for schedule(static)
        for (int mm = 0; mm < M; ++mm) for (int n = n_full; n < N; ++n) {
            float acc = 0.0f; const float* a = A + (long)mm * lda;
            for (int k = 0; k < K; ++k) acc = fmaf(a[k], Wt[(long)k * ldw + n], acc);
            C[(long)mm * ldc + n] = acc;
        }
    }
}

static void round_f16_inplace(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = skw_round_f16(x[i]);
}

/* ggml_compute_forward_norm_f32 followed by ggml_mul(w) and ggml_add(b); out may alias nothing */
static void layer_norm(const float* x, int rows, int d, const ln_t* ln, float* y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * d; float* yr = y + (size_t)r * d;
        double sum = 0.0; for (int i = 0; i < d; ++i) sum += (double)xr[i];
        float mean = (float)(sum / d);
        double sum2 = 0.0; for (int i = 0; i < d; ++i) { float v = xr[i] - mean; yr[i] = v; sum2 += (double)(v * v); }
        float variance = (float)(sum2 / d);
        const float scale = 1.0f / sqrtf(variance + 1e-5f);
        for (int i = 0; i < d; ++i) { float v = yr[i] * scale; v = v * ln->w[i]; yr[i] = v + ln->b[i]; }
    }
}

/* y = round16?( A16 . W + b ) helper: A (rows x n_in, f16-valued), result rows x n_out f32 with bias */
/* Decode-step contraction (D3'): the K axis in four contiguous segments, each a k-ascending fma chain from zero, the four partial
 * sums added in ascending segment order: ((s0 + s1) + s2) + s3.  Every mul_mat of the one-token decoder graph is defined this way
 * (K % 128 == 0 in every Whisper geometry; otherwise the single chain) — four waves share a contraction on the GPU instead of one. */
static void gemm_chain_seg4(const float* A, long lda, int M, const float* Wt, long ldw, int N, int K, float* C, long ldc) {
    if (K % 128) { gemm_chain(A, lda, M, Wt, ldw, N, K, C, ldc); return; }
    const int Kq = K / 4; float* t = xmalloc_f((size_t)4 * M * N);
    for (int s = 0; s < 4; ++s) gemm_chain(A + (long)s * Kq, lda, M, Wt + (long)s * Kq * ldw, ldw, N, Kq, t + (size_t)s * M * N, N);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        const size_t i = (size_t)m * N + n; float v = t[i] + t[(size_t)M * N + i]; v = v + t[(size_t)2 * M * N + i]; v = v + t[(size_t)3 * M * N + i];
        C[(long)m * ldc + n] = v;
    }
    free(t);
}
/* ggml's quantised mul_mat (include/skw_ggml_quant.h (a)): rows of A (f32, NOT rounded to f16) -> q8 blocks, block-ascending chain of integer dots */
static void linear_q8_seg(const float* A, long lda, int rows, const lin_t* L, float* out, long ldo, int nseg) {
    const int K = L->n_in, nb = K / 32, N = L->n_out, form = skw_ggml_dot_form(L->qtype); const int bps = nb / nseg;   /* blocks per segment */
    int8_t* qa = (int8_t*)malloc((size_t)rows * K); float* da = xmalloc_f((size_t)rows * nb); float* sa = xmalloc_f((size_t)rows * nb);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) for (int b = 0; b < nb; ++b) skw_ggml_quantize_q8_block(A + (long)r * lda + b * 32, qa + ((size_t)r * nb + b) * 32,
        &da[(size_t)r * nb + b], &sa[(size_t)r * nb + b]);
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < rows; ++r) for (int n = 0; n < N; ++n) {
        const int8_t* x = qa + (size_t)r * K; const int8_t* w = L->qw + (size_t)n * K;
        float total = 0.0f;
        for (int sg = 0; sg < nseg; ++sg) {       /* nseg == 1: ggml's single block-ascending sum; 4: the decoder's segmented form (D3'), added in ascending order */
            float sumf = 0.0f;
            for (int b = sg * bps; b < (sg + 1) * bps; ++b) {
                int sumi = 0; for (int j = 0; j < 32; ++j) sumi += (int)w[b * 32 + j] * (int)x[b * 32 + j];
                sumf = skw_ggml_block_dot(form, sumf, sumi, L->qd[(size_t)n * nb + b], L->qm[(size_t)n * nb + b], da[(size_t)r * nb + b], sa[(size_t)r * nb + b]);
            }
            total = sg == 0 ? sumf : total + sumf;
        }
        out[(long)r * ldo + n] = total;
    }
    free(qa); free(da); free(sa);
}
static void linear_q8(const float* A, long lda, int rows, const lin_t* L, float* out, long ldo) { linear_q8_seg(A, lda, rows, L, out, ldo, 1); }
/* test hook: out [rows][n_out] = ggml's quantised mul_mat of A [rows][n_in] f32 with n_out * n_in / 32 blocks of the given type */
int skwo_debug_linear_q8(int type, const uint8_t* blocks, int n_out, int n_in, const float* A, int rows, float* out) {
    const size_t bb = skw_ggml_block_bytes(type); if (!bb || n_in % 32) return -1;
    lin_t L; memset(&L, 0, sizeof L); L.n_in = n_in; L.n_out = n_out; L.qtype = type; const int nb = n_in / 32;
    L.qw = (int8_t*)malloc((size_t)n_out * n_in); L.qd = xmalloc_f((size_t)n_out * nb); L.qm = xmalloc_f((size_t)n_out * nb);
    for (size_t i = 0; i < (size_t)n_out * nb; ++i) skw_ggml_unpack_block(type, blocks + i * bb, L.qw + i * 32, &L.qd[i], &L.qm[i]);
    linear_q8(A, n_in, rows, &L, out, n_out);
    free(L.qw); free(L.qd); free(L.qm); return 0;
}
static void linear(const float* A, long lda, int rows, const lin_t* L, float* out, long ldo) {
    if (L->qw) linear_q8(A, lda, rows, L, out, ldo);
    else gemm_chain(A, lda, rows, L->wt, L->n_out, L->n_out, L->n_in, out, ldo);
    if (L->b) {
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; ++r) { float* o = out + (long)r * ldo; for (int i = 0; i < L->n_out; ++i) o[i] = o[i] + L->b[i]; }
    }
}

/* the decoder's projections: segmented contraction (f16 weights) or ggml's q8 arithmetic (quantised files), then the bias */
static void linear_dec(const float* A, long lda, int rows, const lin_t* L, float* out, long ldo) {
    if (L->qw) {   /* quantised weights: the block sums of the decoder's products are segmented the same way (four contiguous runs of blocks) */
        linear_q8_seg(A, lda, rows, L, out, ldo, (L->n_in % 128) ? 1 : 4);
        if (L->b) for (int r = 0; r < rows; ++r) { float* o = out + (long)r * ldo; for (int i = 0; i < L->n_out; ++i) o[i] = o[i] + L->b[i]; }
        return;
    }
    gemm_chain_seg4(A, lda, rows, L->wt, L->n_out, L->n_out, L->n_in, out, ldo);
    if (L->b) for (int r = 0; r < rows; ++r) { float* o = out + (long)r * ldo; for (int i = 0; i < L->n_out; ++i) o[i] = o[i] + L->b[i]; }
}

/* ggml_soft_max_ext row: wp = s*scale; max; p = expf(wp-max); sum (double); p *= (float)(1/sum) */
static float *g_dbg_max = NULL, *g_dbg_inv = NULL; /* [row] when taps are on */
static void softmax_row2(float* s, int n, float scale, float* omax, float* oinv);
static void softmax_row(float* s, int n, float scale) { softmax_row2(s, n, scale, NULL, NULL); }
static void softmax_row2(float* s, int n, float scale, float* omax, float* oinv) {
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) { s[i] = s[i] * scale; if (s[i] > mx) mx = s[i]; }
    double sum = 0.0;
    for (int i = 0; i < n; ++i) { float v = skw_expf(s[i] - mx); s[i] = v; sum += (double)v; }
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < n; ++i) s[i] = s[i] * inv;
    if (omax) *omax = mx;
    if (oinv) *oinv = inv;
}

/* --------------------------------------------- K2: conv stem (+pos emb) */
/* whisper_build_graph_conv: conv1d(k3,p1)+bias+GELU, conv1d(k3,s2,p1)+bias+GELU; then (in the encoder graph)
 * cur = e_pe + transpose(cur). x0: [n_ctx][d]. */
static void conv_stem(const skwo_model* m, const float* mel, int n_len, int seek, float* x0) {
    const int n_ctx = m->hp.n_audio_ctx, T = 2 * n_ctx, n_mel = m->hp.n_mels, d = m->hp.n_audio_state;
    /* im2col source, time-major with one zero row of padding each side; values rounded to f16 (ggml_im2col dst type f16) */
    float* in1 = (float*)calloc((size_t)(T + 2) * n_mel, sizeof(float));
    int i0 = seek < n_len ? seek : n_len, i1 = (seek + T) < n_len ? (seek + T) : n_len;
    for (int j = 0; j < n_mel; ++j) for (int i = i0; i < i1; ++i) in1[(size_t)(i - i0 + 1) * n_mel + j] = skw_round_f16(mel[(size_t)j * n_len + i]);
    float* h1 = xmalloc_f((size_t)(T + 2) * d);
    memset(h1, 0, sizeof(float) * d); memset(h1 + (size_t)(T + 1) * d, 0, sizeof(float) * d);
    /* conv1: out[t] = chain over k = tap*n_mel + c of in[t-1+tap][c] * w[oc][c][tap]  -> rows are overlapping windows (lda = n_mel) */
    linear(in1, n_mel, T, &m->conv1, h1 + d, d);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)T * d; ++i) h1[d + i] = skw_round_f16(skw_gelu_lookup(h1[d + i], m->gelu_tab)); /* gelu, then im2col(f16) of conv2 */
    /* conv2 stride 2: out[t] reads rows 2t-1..2t+1 => window starts at padded row 2t, lda = 2*d, K = 3*d */
    linear(h1, 2 * d, n_ctx, &m->conv2, x0, d);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)n_ctx * d; ++i) { float g = skw_gelu_lookup(x0[i], m->gelu_tab); x0[i] = m->e_pe[i] + g; }
    free(in1); free(h1);
}

int skwo_conv_stem(const skwo_model* m, const float* mel, int n_len, int seek, int n_threads, float* x0) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    conv_stem(m, mel, n_len, seek, x0); return 0;
}

/* ------------------------------------------------ K3-K6: encoder + cross */
int skwo_encode(const skwo_model* m, const float* mel, int n_len, int seek, int n_threads, float* enc_out, float* cross_k, float* cross_v) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int n_ctx = m->hp.n_audio_ctx, d = m->hp.n_audio_state, nh = m->hp.n_audio_head, dh = d / nh;
    float* x = xmalloc_f((size_t)n_ctx * d); float* y = xmalloc_f((size_t)n_ctx * d);
    float* q = xmalloc_f((size_t)n_ctx * d); float* kk = xmalloc_f((size_t)n_ctx * d); float* v = xmalloc_f((size_t)n_ctx * d);
    float* att = xmalloc_f((size_t)n_ctx * d); float* hbuf = xmalloc_f((size_t)n_ctx * 4 * d);
    float* kt = xmalloc_f((size_t)dh * n_ctx); float* S = xmalloc_f((size_t)n_ctx * n_ctx); float* vh = xmalloc_f((size_t)n_ctx * dh); float* oh = xmalloc_f((size_t)n_ctx * dh);
    conv_stem(m, mel, n_len, seek, x);
    const float KQscale = 1.0f / sqrtf((float)dh);
    for (int l = 0; l < m->hp.n_audio_layer; ++l) {
        const enc_layer_t* L = &m->enc[l];
        layer_norm(x, n_ctx, d, &L->attn_ln, y); if (!m->quant) round_f16_inplace(y, (size_t)n_ctx * d);   /* mul_mat's src1: -> f16 for f16 weights, -> q8 blocks (inside linear) for quantised ones */
        if (l == 0) tap("l0.ln1", y, (size_t)n_ctx * d);
        linear(y, d, n_ctx, &L->q, q, d); linear(y, d, n_ctx, &L->k, kk, d); linear(y, d, n_ctx, &L->v, v, d);
        round_f16_inplace(q, (size_t)n_ctx * d);   /* src1 of mul_mat(K,Q) -> f16 */
        round_f16_inplace(kk, (size_t)n_ctx * d);  /* ggml_cast(K, f16) */
        round_f16_inplace(v, (size_t)n_ctx * d);   /* ggml_cast(V, f16) */
        if (l == 0) { tap("l0.q", q, (size_t)n_ctx * d); tap("l0.k", kk, (size_t)n_ctx * d); tap("l0.v", v, (size_t)n_ctx * d); }
        if (l == 0 && g_taps_on) { g_dbg_max = (float*)calloc((size_t)nh * n_ctx, 4); g_dbg_inv = (float*)calloc((size_t)nh * n_ctx, 4); }
        for (int h = 0; h < nh; ++h) {
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n_ctx; ++j) for (int c = 0; c < dh; ++c) { kt[(size_t)c * n_ctx + j] = kk[(size_t)j * d + h * dh + c]; vh[(size_t)j * dh + c] = v[(size_t)j * d + h * dh + c]; }
            gemm_chain(q + h * dh, d, n_ctx, kt, n_ctx, n_ctx, dh, S, n_ctx);
            float* spd = NULL;
            if (l == 0 && h == 0 && g_taps_on) { int Tp = (n_ctx + 31) & ~31;
            spd = (float*)calloc((size_t)64 * Tp, 4); for (int i = 0; i < 32; ++i) for (int j = 0; j < n_ctx; ++j) spd[(size_t)i * Tp + j] = S[(size_t)i * n_ctx + j] * KQscale;
            for (int i = 0; i < 32; ++i) for (int j = n_ctx; j < Tp; ++j) spd[(size_t)i * Tp + j] = -INFINITY; }
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n_ctx; ++i) { float* s = S + (size_t)i * n_ctx;
            float mx, iv; softmax_row2(s, n_ctx, KQscale, &mx, &iv); if (g_dbg_max) { g_dbg_max[(size_t)h * n_ctx + i] = mx;
            g_dbg_inv[(size_t)h * n_ctx + i] = iv; } for (int j = 0; j < n_ctx; ++j) s[j] = skw_round_f16(s[j]); }
            if (spd) { int Tp = (n_ctx + 31) & ~31; for (int i = 0; i < 32; ++i) for (int j = 0; j < n_ctx; ++j) spd[(size_t)(32 + i) * Tp + j] = S[(size_t)i * n_ctx + j];
            tap("l0.SP", spd, (size_t)64 * Tp); free(spd); }
            gemm_chain(S, n_ctx, n_ctx, vh, dh, dh, n_ctx, oh, dh);
            if (l == 0 && g_taps_on) { static float* a32 = NULL; if (h == 0) a32 = (float*)calloc((size_t)n_ctx * d, 4);
            for (int i = 0; i < n_ctx; ++i) for (int c = 0; c < dh; ++c) a32[(size_t)i * d + h * dh + c] = oh[(size_t)i * dh + c];
            if (h == nh - 1) { tap("l0.att32", a32, (size_t)n_ctx * d); free(a32); } }
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n_ctx; ++i) for (int c = 0; c < dh; ++c) att[(size_t)i * d + h * dh + c] = m->quant ? oh[(size_t)i * dh + c] : skw_round_f16(oh[(size_t)i * dh + c]);
        }
        if (l == 0) tap("l0.att", att, (size_t)n_ctx * d);
        if (g_dbg_max) { tap("l0.rmax", g_dbg_max, (size_t)nh * n_ctx); tap("l0.rinv", g_dbg_inv, (size_t)nh * n_ctx); free(g_dbg_max); free(g_dbg_inv); g_dbg_max = g_dbg_inv = NULL; }
        linear(att, d, n_ctx, &L->o, y, d);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)n_ctx * d; ++i) x[i] = y[i] + x[i];
        if (l == 0) tap("l0.x1", x, (size_t)n_ctx * d);
        layer_norm(x, n_ctx, d, &L->mlp_ln, y); if (!m->quant) round_f16_inplace(y, (size_t)n_ctx * d);
        if (l == 0) tap("l0.ln2", y, (size_t)n_ctx * d);
        linear(y, d, n_ctx, &L->fc1, hbuf, 4 * d);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)n_ctx * 4 * d; ++i) { const float g = skw_gelu_lookup(hbuf[i], m->gelu_tab); hbuf[i] = m->quant ? g : skw_round_f16(g); }
        if (l == 0) tap("l0.h", hbuf, (size_t)n_ctx * 4 * d);
        linear(hbuf, 4 * d, n_ctx, &L->fc2, y, d);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)n_ctx * d; ++i) x[i] = y[i] + x[i];
        if (l == 0) tap("l0.x2", x, (size_t)n_ctx * d);
    }
    layer_norm(x, n_ctx, d, &m->ln_post, enc_out);
    if (cross_k && cross_v) { /* whisper_build_graph_cross */
        const int dtxt = m->hp.n_text_state; const float Kscale = (float)pow((double)((float)dtxt / m->hp.n_text_head), -0.25);
        memcpy(y, enc_out, sizeof(float) * (size_t)n_ctx * d); if (!m->quant) round_f16_inplace(y, (size_t)n_ctx * d);
        for (int l = 0; l < m->hp.n_text_layer; ++l) {
            float* ck = cross_k + (size_t)l * n_ctx * dtxt; float* cv = cross_v + (size_t)l * n_ctx * dtxt;
            linear(y, d, n_ctx, &m->dec[l].ck, ck, dtxt); linear(y, d, n_ctx, &m->dec[l].cv, cv, dtxt);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < (size_t)n_ctx * dtxt; ++i) { ck[i] = skw_round_f16(ck[i] * Kscale); cv[i] = skw_round_f16(cv[i]); }
        }
    }
    free(x); free(y); free(q); free(kk); free(v); free(att); free(hbuf); free(kt); free(S); free(vh); free(oh);
    return 0;
}

/* ---------------------------------------------------- K7-K10: decoder */
struct skwo_dec {
    const skwo_model* m; const float *cross_k, *cross_v;
    float *self_k, *self_v; /* [layer][n_text_ctx][d] f16-valued */
    float* kxt; /* per layer/head transposed cross K: [layer][head][dh][n_ctx] for vectorised chains */
};
skwo_dec* skwo_dec_new(const skwo_model* m, const float* cross_k, const float* cross_v) {
    skwo_dec* s = (skwo_dec*)calloc(1, sizeof *s); s->m = m; s->cross_k = cross_k; s->cross_v = cross_v;
    const int d = m->hp.n_text_state, nl = m->hp.n_text_layer, nc = m->hp.n_audio_ctx, nh = m->hp.n_text_head, dh = d / nh;
    s->self_k = xmalloc_f((size_t)nl * m->hp.n_text_ctx * d); s->self_v = xmalloc_f((size_t)nl * m->hp.n_text_ctx * d);
    s->kxt = xmalloc_f((size_t)nl * d * nc);
#pragma omp parallel for collapse(2) schedule(static)
    for (int l = 0; l < nl; ++l) for (int h = 0; h < nh; ++h) for (int c = 0; c < dh; ++c) for (int j = 0; j < nc; ++j)
        s->kxt[(((size_t)l * nh + h) * dh + c) * nc + j] = cross_k[((size_t)l * nc + j) * d + h * dh + c];
    return s;
}
void skwo_dec_free(skwo_dec* s) { if (!s) return; free(s->self_k); free(s->self_v); free(s->kxt); free(s); }
void skwo_dec_reset(skwo_dec* s) { (void)s; }

/* one token at absolute position pos; fills logits if non-NULL */
static void dec_one(skwo_dec* s, int token, int pos, float* logits) {
    const skwo_model* m = s->m; const int d = m->hp.n_text_state, nh = m->hp.n_text_head, dh = d / nh, nc = m->hp.n_audio_ctx, ntc = m->hp.n_text_ctx;
    const float KQscale = (float)pow((double)((float)d / nh), -0.25);
    float x[2048], y[2048], q[2048], kv[2048], att[2048], sc[2048]; float* hb = (float*)alloca(sizeof(float) * 4 * d);
    for (int i = 0; i < d; ++i) x[i] = m->d_te[(size_t)token * d + i] + m->d_pe[(size_t)pos * d + i];
    for (int l = 0; l < m->hp.n_text_layer; ++l) {
        const dec_layer_t* L = &m->dec[l];
        float* Kc = s->self_k + (size_t)l * ntc * d; float* Vc = s->self_v + (size_t)l * ntc * d;
        /* self-attention */
        layer_norm(x, 1, d, &L->attn_ln, y); if (!m->quant) for (int i = 0; i < d; ++i) y[i] = skw_round_f16(y[i]);
        linear_dec(y, d, 1, &L->q, q, d); for (int i = 0; i < d; ++i) q[i] = skw_round_f16(q[i] * KQscale);
        linear_dec(y, d, 1, &L->k, kv, d); for (int i = 0; i < d; ++i) Kc[(size_t)pos * d + i] = skw_round_f16(kv[i] * KQscale);
        linear_dec(y, d, 1, &L->v, kv, d); for (int i = 0; i < d; ++i) Vc[(size_t)pos * d + i] = skw_round_f16(kv[i]);
        const int n_kv = pos + 1;
        for (int h = 0; h < nh; ++h) {
            for (int j = 0; j < n_kv; ++j) { float a = 0.0f; const float* kr = Kc + (size_t)j * d + h * dh; for (int c = 0; c < dh; ++c) a = fmaf(q[h * dh + c], kr[c], a); sc[j] = a; }
            softmax_row(sc, n_kv, 1.0f);
            for (int j = 0; j < n_kv; ++j) sc[j] = skw_round_f16(sc[j]);
            for (int c = 0; c < dh; ++c) { float a = 0.0f; for (int j = 0; j < n_kv; ++j) a = fmaf(sc[j], Vc[(size_t)j * d + h * dh + c], a); att[h * dh + c] = m->quant ? a : skw_round_f16(a); }
        }
        linear_dec(att, d, 1, &L->o, y, d); for (int i = 0; i < d; ++i) x[i] = y[i] + x[i];
        /* cross-attention */
        layer_norm(x, 1, d, &L->cross_ln, y); if (!m->quant) for (int i = 0; i < d; ++i) y[i] = skw_round_f16(y[i]);
        linear_dec(y, d, 1, &L->cq, q, d); for (int i = 0; i < d; ++i) q[i] = skw_round_f16(q[i] * KQscale);
        const float* Vx = s->cross_v + (size_t)l * nc * d;
        for (int h = 0; h < nh; ++h) {
            float scl[1536];
            const float* kt = s->kxt + ((size_t)l * nh + h) * dh * nc;
            gemm_chain(q + h * dh, d, 1, kt, nc, nc, dh, scl, nc);
            softmax_row(scl, nc, 1.0f);
            for (int j = 0; j < nc; ++j) scl[j] = skw_round_f16(scl[j]);
            for (int c = 0; c < dh; ++c) { float a = 0.0f; for (int j = 0; j < nc; ++j) a = fmaf(scl[j], Vx[(size_t)j * d + h * dh + c], a); att[h * dh + c] = m->quant ? a : skw_round_f16(a); }
        }
        linear_dec(att, d, 1, &L->co, y, d); for (int i = 0; i < d; ++i) x[i] = y[i] + x[i];
        /* mlp */
        layer_norm(x, 1, d, &L->mlp_ln, y); if (!m->quant) for (int i = 0; i < d; ++i) y[i] = skw_round_f16(y[i]);
        linear_dec(y, d, 1, &L->fc1, hb, 4 * d); for (int i = 0; i < 4 * d; ++i) { const float g = skw_gelu_lookup(hb[i], m->gelu_tab); hb[i] = m->quant ? g : skw_round_f16(g); }
        linear_dec(hb, 4 * d, 1, &L->fc2, y, d); for (int i = 0; i < d; ++i) x[i] = y[i] + x[i];
    }
    if (logits) {
        layer_norm(x, 1, d, &m->d_ln, y); if (!m->quant) for (int i = 0; i < d; ++i) y[i] = skw_round_f16(y[i]);
        if (m->d_te_lin.qw) linear_q8_seg(y, d, 1, &m->d_te_lin, logits, m->hp.n_vocab, (d % 128) ? 1 : 4);
        else gemm_chain_seg4(y, d, 1, m->d_te_lin.wt, m->d_te_lin.n_out, m->hp.n_vocab, d, logits, m->hp.n_vocab);
    }
}
int skwo_dec_step(skwo_dec* s, const int32_t* tokens, int n_tokens, int n_past, int n_threads, float* logits) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    if (s->m->hp.n_text_state > 2048 || s->m->hp.n_audio_ctx > 1536) return -1;
    if (n_past + n_tokens > s->m->hp.n_text_ctx) return -2;
    for (int i = 0; i < n_tokens; ++i) dec_one(s, tokens[i], n_past + i, i == n_tokens - 1 ? logits : NULL);
    return 0;
}

/* ------------------------------------------- K11: whisper_process_logits */
typedef struct {
    skwo_token* tokens; int n_tokens, cap;   /* sequence.tokens */
    int result_len; int seek_delta; int has_ts, failed, completed;
    double sum_logprobs_all, sum_logprobs, avg_logprobs, entropy, score;
    float *logits, *logprobs, *probs;
    float min_margin;
} decoder_t;

static void compute_logprobs(const float* logits, int n, float* logprobs) {
    float logit_max = -INFINITY; for (int i = 0; i < n; ++i) if (logits[i] > logit_max) logit_max = logits[i];
    double acc = 0.0; /* DEVIATION D1: whisper.cpp sums in f32 in token order; f64 makes the sum order-insensitive */
    for (int i = 0; i < n; ++i) if (logits[i] > -INFINITY) acc += (double)skw_expf(logits[i] - logit_max);
    float logsumexp = skw_logf((float)acc) + logit_max;
    for (int i = 0; i < n; ++i) logprobs[i] = (logits[i] > -INFINITY) ? logits[i] - logsumexp : -INFINITY;
}

static int g_dbg_no_mass_rule = 0;   /* test hook only (skwo_debug_process_logits flag 1): leaves the timestamp-mass rule out so the index rules can be compared on their own */
static void process_logits(const skwo_model* m, const skwo_params* p, decoder_t* dc, const float* raw_logits, float* no_speech_prob, float temperature) {
    const int n_logits = m->hp.n_vocab; float* logits = dc->logits; float* logprobs = dc->logprobs; float* probs = dc->probs;
    const int is_initial = dc->n_tokens == 0;
    memcpy(logits, raw_logits, sizeof(float) * n_logits);
    if (is_initial) { /* no-speech probability from the unfiltered distribution */
        compute_logprobs(logits, n_logits, logprobs);
        *no_speech_prob = skw_expf(logprobs[m->tok_nosp]);
    }
    if (temperature > 0.0f) for (int i = 0; i < n_logits; ++i) logits[i] /= temperature;   /* whisper_process_logits: before any filter */
    if (p->suppress_blank && is_initial) { logits[m->tok_eot] = -INFINITY; if (m->tok_space >= 0) logits[m->tok_space] = -INFINITY; }
    logits[m->tok_not] = -INFINITY;
    if (p->no_timestamps) for (int i = m->tok_beg; i < n_logits; ++i) logits[i] = -INFINITY;
    logits[m->tok_sot] = -INFINITY; logits[m->tok_nosp] = -INFINITY;
    logits[m->tok_solm] = -INFINITY; /* tdrz disabled */
    logits[m->tok_translate] = -INFINITY; logits[m->tok_transcribe] = -INFINITY; logits[m->tok_prev] = -INFINITY;
    { int n_lang = N_LANG + (n_logits - 51865 > 0 ? n_logits - 51865 : 0); for (int i = 0; i < n_lang; ++i) logits[m->tok_sot + 1 + i] = -INFINITY; }
    if (p->suppress_nst) {
        for (int i = 0; i < m->n_nst; ++i) logits[m->nst_ids[i]] = -INFINITY;
        if (m->tok_sp_dash >= 0) logits[m->tok_sp_dash] = -INFINITY;
        if (m->tok_sp_quote >= 0) logits[m->tok_sp_quote] = -INFINITY;
    }
    { /* timestamps have to appear in pairs, except directly before EOT */
        const int last_was_timestamp = dc->n_tokens > 0 && dc->tokens[dc->n_tokens - 1].id >= m->tok_beg;
        const int penultimate_was_timestamp = dc->n_tokens < 2 || dc->tokens[dc->n_tokens - 2].id >= m->tok_beg;
        if (last_was_timestamp) {
            if (penultimate_was_timestamp) { for (int i = m->tok_beg; i < n_logits; ++i) logits[i] = -INFINITY; }
            else { for (int i = 0; i < m->tok_eot; ++i) logits[i] = -INFINITY; }
        }
    }
    if (is_initial && p->max_initial_ts > 0.0f) {
        const float precision = (float)WHISPER_CHUNK_SIZE / m->hp.n_audio_ctx;
        const int tid0 = (int)roundf(p->max_initial_ts / precision);
        for (int i = m->tok_beg + tid0 + 1; i < n_logits; ++i) logits[i] = -INFINITY;
    }
    if (dc->has_ts) { const int tid0 = dc->seek_delta / 2; for (int i = m->tok_beg; i < m->tok_beg + tid0 && i < n_logits; ++i) logits[i] = -INFINITY; }
    compute_logprobs(logits, n_logits, logprobs);
    { /* if sum of probability over timestamps is above any other token, sample timestamp */
        float timestamp_logprob = -INFINITY;
        {
            float logprob_max = -INFINITY; for (int i = m->tok_beg; i < n_logits; ++i) if (logprobs[i] > logprob_max) logprob_max = logprobs[i];
            double acc = 0.0; for (int i = m->tok_beg; i < n_logits; ++i) if (logprobs[i] > -INFINITY) acc += (double)skw_expf(logprobs[i] - logprob_max);
            if (acc > 0.0) timestamp_logprob = skw_logf((float)acc) + logprob_max;
        }
        float max_text = -INFINITY; for (int i = 0; i < m->tok_beg; ++i) if (logprobs[i] > max_text) max_text = logprobs[i];
        if (timestamp_logprob > max_text && !g_dbg_no_mass_rule) for (int i = 0; i < m->tok_beg; ++i) { logits[i] = -INFINITY; logprobs[i] = -INFINITY; }
    }
    for (int i = 0; i < n_logits; ++i) probs[i] = (logits[i] == -INFINITY) ? 0.0f : skw_expf(logprobs[i]);
}

/* whisper_sample_token(best = true) */
static skwo_token sample_best(const skwo_model* m, decoder_t* dc) {
    skwo_token r = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, INFINITY}; const int n = m->hp.n_vocab; const float* probs = dc->probs;
    { double sum_ts = 0.0, max_ts = 0.0; for (int i = m->tok_beg; i < n; ++i) { sum_ts += probs[i];
    if (max_ts < probs[i]) { max_ts = probs[i]; r.tid = i; } } r.pt = (float)(max_ts / (sum_ts + 1e-10)); r.ptsum = (float)sum_ts; }
    for (int i = 0; i < n; ++i) if (r.p < probs[i]) { r.id = i; r.p = probs[i]; r.plog = dc->logprobs[i]; }
    if (r.id >= m->tok_beg) { r.tid = r.id; r.pt = r.p; }
    { /* diagnostics: margin between the two largest admissible logits */
        float a = -INFINITY, b = -INFINITY; for (int i = 0; i < n; ++i) { float v = dc->logits[i]; if (v > a) { b = a; a = v; } else if (v > b) b = v; }
        if (b > -INFINITY) { r.margin = a - b; if (a - b < dc->min_margin) dc->min_margin = a - b; }
    }
    return r;
}

/* std::mt19937 as libstdc++ implements it (the generator whisper.cpp keeps per decoder, seeded with 0) */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t* g, uint32_t seed) {
    g->mt[0] = seed; for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
/* std::generate_canonical<double, 53>(mt19937): two draws, low word first (libstdc++ bits/random.tcc) */
static double mt_canonical(mt19937_t* g) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) { sum += (double)mt_next(g) * tmp; tmp *= 4294967296.0; }
    double r = sum / tmp;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}
/* std::discrete_distribution<>(probs.begin(), probs.end())(rng): weights to double, divided by their sequential sum,
 * sequential partial sums with the last forced to 1.0, lower_bound of one canonical draw */
static int discrete_draw(const float* probs, int n, mt19937_t* g) {
    double sum = 0.0; for (int i = 0; i < n; ++i) sum += (double)probs[i];
    const double u = mt_canonical(g);
    double cp = 0.0;
    for (int i = 0; i < n; ++i) { cp += (double)probs[i] / sum; if (i == n - 1) cp = 1.0; if (!(cp < u)) return i; }
    return n - 1;
}
int skwo_discrete_draw(const float* probs, int n, uint32_t seed, int n_draws, int32_t* out) {   /* test hook (pinned against libstdc++ in tests/) */
    mt19937_t g; mt_seed(&g, seed); for (int k = 0; k < n_draws; ++k) out[k] = discrete_draw(probs, n, &g); return 0;
}
/* whisper_sample_token(best = false) */
static skwo_token sample_dist(const skwo_model* m, decoder_t* dc, mt19937_t* rng) {
    skwo_token r = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, INFINITY}; const int n = m->hp.n_vocab; const float* probs = dc->probs;
    { double sum_ts = 0.0, max_ts = 0.0; for (int i = m->tok_beg; i < n; ++i) { sum_ts += probs[i];
    if (max_ts < probs[i]) { max_ts = probs[i]; r.tid = i; } } r.pt = (float)(max_ts / (sum_ts + 1e-10)); r.ptsum = (float)sum_ts; }
    r.id = discrete_draw(probs, n, rng); r.p = probs[r.id]; r.plog = dc->logprobs[r.id];
    if (r.id >= m->tok_beg) { r.tid = r.id; r.pt = r.p; }
    return r;
}

/* test hook: ONE call of whisper_process_logits + whisper_sample_token(best = true) on caller-supplied logits, for a decoder whose tokens
 * sampled so far in this window are hist[0 .. n_hist).  has_ts / seek_delta / result_len follow from the history by the update rule of
 * whisper_full_with_state's token loop (above), so the history must be one that loop can produce (timestamps never decrease).
 * flags bit 0: leave the timestamp-mass rule out.  out_logits: the filtered logits (-inf = suppressed).  Pinned against an independent
 * implementation of the same rules (transformers' Whisper logits processors) by tests/test_cpu_logit_rules.py. */
int skwo_debug_process_logits(const skwo_model* m, const skwo_params* p, const int32_t* hist, int n_hist, const float* raw_logits, float temperature, int flags,
                              float* out_logits, float* out_logprobs, skwo_token* chosen, float* no_speech_prob) {
    const int NV = m->hp.n_vocab;
    decoder_t dc; memset(&dc, 0, sizeof dc);
    dc.logits = xmalloc_f(NV); dc.logprobs = xmalloc_f(NV); dc.probs = xmalloc_f(NV); dc.cap = n_hist + 1; dc.tokens = (skwo_token*)calloc(dc.cap, sizeof(skwo_token)); dc.min_margin = INFINITY;
    dc.seek_delta = 100 * WHISPER_CHUNK_SIZE; dc.has_ts = 0;
    for (int i = 0; i < n_hist; ++i) {
        dc.tokens[i].id = hist[i]; dc.n_tokens = i + 1;
        if (hist[i] > m->tok_beg) {
            const int sd = 2 * (hist[i] - m->tok_beg);
            if (dc.has_ts && dc.seek_delta > sd && dc.result_len < i) { free(dc.logits); free(dc.logprobs); free(dc.probs); free(dc.tokens); return -1; }   /* the loop would have failed here */
            dc.seek_delta = sd; dc.result_len = i + 1; dc.has_ts = 1;
        }
    }
    float nsp = 0.0f;
    g_dbg_no_mass_rule = flags & 1;
    process_logits(m, p, &dc, raw_logits, &nsp, temperature);
    g_dbg_no_mass_rule = 0;
    if (out_logits) memcpy(out_logits, dc.logits, sizeof(float) * NV);
    if (out_logprobs) memcpy(out_logprobs, dc.logprobs, sizeof(float) * NV);
    if (chosen) *chosen = sample_best(m, &dc);
    if (no_speech_prob) *no_speech_prob = nsp;
    free(dc.logits); free(dc.logprobs); free(dc.probs); free(dc.tokens);
    return 0;
}
/* the ids the static suppression rules name, for the checker's side of the comparison: kind 0 = specials (not, sot, nosp, solm, translate, transcribe, prev, languages),
 * 1 = the non-speech list (suppress_nst), 2 = the blank rule's pair (eot, " ").  Returns the count (ids may be NULL). */
int skwo_debug_rule_ids(const skwo_model* m, int kind, int32_t* ids, int cap) {
    int n = 0;
#define PUSH(x) do { if ((x) >= 0) { if (ids && n < cap) ids[n] = (x); n++; } } while (0)
    if (kind == 0) {
        PUSH(m->tok_not); PUSH(m->tok_sot); PUSH(m->tok_nosp); PUSH(m->tok_solm); PUSH(m->tok_translate); PUSH(m->tok_transcribe); PUSH(m->tok_prev);
        const int n_lang = N_LANG + (m->hp.n_vocab - 51865 > 0 ? m->hp.n_vocab - 51865 : 0); for (int i = 0; i < n_lang; ++i) PUSH(m->tok_sot + 1 + i);
    } else if (kind == 1) {
        for (int i = 0; i < m->n_nst; ++i) PUSH(m->nst_ids[i]);
        PUSH(m->tok_sp_dash); PUSH(m->tok_sp_quote);
    } else if (kind == 2) { PUSH(m->tok_eot); PUSH(m->tok_space); }
    else if (kind == 3) { PUSH(m->tok_eot); PUSH(m->tok_not); PUSH(m->tok_beg); PUSH(m->tok_nosp); PUSH(m->tok_sot); }
#undef PUSH
    return n;
}

static void sequence_score(decoder_t* dc) {
    if (dc->result_len == 0) return;
    double result = 0.0; for (int i = 0; i < dc->result_len; ++i) result += dc->tokens[i].plog;
    dc->sum_logprobs = result; dc->avg_logprobs = result / dc->result_len; dc->score = result / (double)dc->result_len; /* length_penalty <= 0 */
    const int n = 32; int cnt = 0; int ids[32], c[32], nu = 0;
    for (int i = dc->result_len - n > 0 ? dc->result_len - n : 0; i < dc->result_len; ++i) {
        int id = dc->tokens[i].id, f = -1;
        for (int u = 0; u < nu; ++u) if (ids[u] == id) f = u;
        if (f < 0) { ids[nu] = id; c[nu] = 1; nu++; } else c[f]++;
        cnt++;
    }
    /* std::map iterates in key order: sort for identical summation order */
    for (int a = 0; a < nu; ++a) for (int b = a + 1; b < nu; ++b) if (ids[b] < ids[a]) { int t = ids[a]; ids[a] = ids[b]; ids[b] = t; t = c[a]; c[a] = c[b]; c[b] = t; }
    double entropy = 0.0; for (int u = 0; u < nu; ++u) { double pp = c[u] / (double)cnt; entropy -= pp * log(pp); }
    dc->entropy = entropy;
}

/* result accumulation */
typedef struct { skwo_segment* seg; int n_seg, cap_seg; skwo_token* tok; int n_tok, cap_tok; char* text; int n_text, cap_text; } acc_t;
static void acc_push_seg(acc_t* a, const skwo_model* m, int64_t t0, int64_t t1, const char* text, int tl, const skwo_token* toks, int i0, int i1) {
    if (a->n_seg == a->cap_seg) { a->cap_seg = a->cap_seg ? a->cap_seg * 2 : 16; a->seg = (skwo_segment*)realloc(a->seg, a->cap_seg * sizeof(skwo_segment)); }
    while (a->n_tok + (i1 - i0) > a->cap_tok) { a->cap_tok = a->cap_tok ? a->cap_tok * 2 : 256; a->tok = (skwo_token*)realloc(a->tok, a->cap_tok * sizeof(skwo_token)); }
    while (a->n_text + tl + 1 > a->cap_text) { a->cap_text = a->cap_text ? a->cap_text * 2 : 1024; a->text = (char*)realloc(a->text, a->cap_text); }
    skwo_segment* s = &a->seg[a->n_seg++]; s->t0 = t0; s->t1 = t1; s->tok_begin = a->n_tok; (void)m;
    for (int i = i0; i < i1; ++i) a->tok[a->n_tok++] = toks[i];
    s->tok_end = a->n_tok; s->text_off = a->n_text; s->text_len = tl; memcpy(a->text + a->n_text, text, tl); a->n_text += tl; a->text[a->n_text] = 0;
}
/* One sampled token's effect on the window's bookkeeping (the body of whisper_full_with_state's token loop, between sampling and the next decoder step; recalled — see the file
 * header): seek_delta / result_len follow the last timestamp token above <|0.00|>, a timestamp that steps BACK fails the pass, the pass completes at <|endoftext|>, at max_tokens,
 * or when the timestamps reach the end of the audio, and fails when the token budget runs out before half a window is covered.  0 = go on, non-zero = leave the loop
 * (dc->completed or dc->failed says which).  Shared by skwo_full and by skwo_debug_window, which tests/test_cpu_segment_rules.py holds against an independent implementation. */
static int token_loop_update(const skwo_model* m, const skwo_params* p, decoder_t* dc, int id, int i, int seek, int seek_end, int n_max) {
    if (id > m->tok_beg) {
        const int seek_delta_new = 2 * (id - m->tok_beg);
        if (dc->has_ts && dc->seek_delta > seek_delta_new && dc->result_len < i) { dc->failed = 1; return 1; }
        dc->seek_delta = seek_delta_new; dc->result_len = i + 1; dc->has_ts = 1;
    }
    if (id == m->tok_eot || (p->max_tokens > 0 && i >= p->max_tokens) || (dc->has_ts && seek + dc->seek_delta + DELTA_MIN >= seek_end)) {
        if (dc->result_len == 0 && !p->no_timestamps) {
            if (seek + dc->seek_delta + DELTA_MIN >= seek_end) dc->result_len = i + 1;
            else { dc->failed = 1; return 1; }
        }
        if (p->single_segment || p->no_timestamps) { dc->result_len = i + 1; dc->seek_delta = 100 * WHISPER_CHUNK_SIZE; }
        dc->completed = 1; return 1;
    }
    if (i == n_max - 1 && (dc->result_len == 0 || dc->seek_delta < 100 * WHISPER_CHUNK_SIZE / 2)) { dc->failed = 1; return 1; }
    return 0;
}
/* The window's output step (whisper_full_with_state after the temperature ladder; recalled): the kept tokens dc->tokens[0 .. n_tokens) are cut into segments at timestamp tokens
 * above <|0.00|> (a run of timestamps closes one segment; a trailing piece of text ends at seek + seek_delta), and the window advances by seek_delta — or, when the tokens end
 * "text, timestamp" (nothing spoken after the last timestamp), by what is left of the chunk.  Returns that advance (10 ms frames).  text: scratch of >= 64 KiB. */
static int window_output(const skwo_model* m, const skwo_params* p, const decoder_t* dc, int seek, int seek_end, int is_no_speech, acc_t* acc, char* text) {
    int seek_delta = dc->seek_delta; const skwo_token* tc = dc->tokens; const int ntc = dc->n_tokens;
    if (ntc > 0 && !is_no_speech) {
        int i0 = 0; int64_t t0 = seek + 2 * (tc[0].tid - m->tok_beg); int tl = 0;
        for (int i = 0; i < ntc; ++i) {
            if (tc[i].id < m->tok_eot) { memcpy(text + tl, m->tok_str[tc[i].id], m->tok_len[tc[i].id]); tl += m->tok_len[tc[i].id]; }
            if (tc[i].id > m->tok_beg && !p->single_segment) {
                const int64_t t1 = seek + 2 * (tc[i].tid - m->tok_beg);
                if (tl > 0) acc_push_seg(acc, m, t0, t1, text, tl, tc, i0, i + 1);
                tl = 0;
                while (i < ntc && tc[i].id > m->tok_beg) i++;
                i--; t0 = t1; i0 = i + 1;
            }
        }
        if (tl > 0) { const int64_t t1 = seek + seek_delta; acc_push_seg(acc, m, t0, t1, text, tl, tc, i0, ntc); }
    }
    const int single_timestamp_ending = ntc > 1 && tc[ntc - 2].id < m->tok_beg && tc[ntc - 1].id > m->tok_beg;
    if (single_timestamp_ending) { int a = seek_end - seek, b = WHISPER_CHUNK_SIZE * 100; seek_delta = a < b ? a : b; }
    return seek_delta;
}

/* test hook (tests/test_cpu_segment_rules.py): ONE window's bookkeeping and output on a caller-supplied stream of sampled token ids — toks[0 .. n) as whisper_full_with_state's token
 * loop would have received them one by one (it stops by itself: <|endoftext|>, a timestamp at the end of the audio, a failure) — for a window at `seek` of audio ending at
 * `seek_end` (10 ms frames).  Out: per segment (t0, t1) in seg_t[2 i ..] and its token ids in seg_tokens[seg_off[i] .. seg_off[i + 1]); the segment count, the advance of seek,
 * how many tokens were kept, how many of toks the loop consumed, and whether the pass failed (then nothing else is meaningful).  Shares token_loop_update and window_output with
 * skwo_full.  Buffers: max_seg segments, n token ids. */
int skwo_debug_window(const skwo_model* m, const skwo_params* p, const int32_t* toks, int n, int seek, int seek_end, int64_t* seg_t, int32_t* seg_off, int32_t* seg_tokens,
                      int max_seg, int* n_seg, int* advance, int* n_kept, int* n_consumed, int* failed) {
    decoder_t dc; memset(&dc, 0, sizeof dc);
    dc.cap = n + 1; dc.tokens = (skwo_token*)calloc(dc.cap, sizeof(skwo_token));
    dc.seek_delta = 100 * WHISPER_CHUNK_SIZE;
    const int n_max = m->hp.n_text_ctx / 2 - 4;
    int i = 0;
    for (; i < n && i < n_max; ++i) {
        skwo_token tk; memset(&tk, 0, sizeof tk); tk.id = toks[i]; tk.tid = toks[i] >= m->tok_beg ? toks[i] : (i > 0 ? dc.tokens[i - 1].tid : m->tok_beg);
        dc.tokens[dc.n_tokens++] = tk;
        if (token_loop_update(m, p, &dc, tk.id, i, seek, seek_end, n_max)) { ++i; break; }
    }
    *n_consumed = i; *failed = dc.failed; *n_seg = 0; *advance = 0; *n_kept = 0;
    if (dc.failed) { free(dc.tokens); return 0; }
    dc.n_tokens = dc.result_len; *n_kept = dc.result_len;
    acc_t acc; memset(&acc, 0, sizeof acc);
    char* text = (char*)malloc(1 << 16);
    *advance = window_output(m, p, &dc, seek, seek_end, 0, &acc, text);
    seg_off[0] = 0;
    for (int s = 0; s < acc.n_seg && s < max_seg; ++s) {
        seg_t[2 * s] = acc.seg[s].t0; seg_t[2 * s + 1] = acc.seg[s].t1; seg_off[s + 1] = acc.seg[s].tok_end;
        for (int k = acc.seg[s].tok_begin; k < acc.seg[s].tok_end; ++k) seg_tokens[k] = acc.tok[k].id;
    }
    *n_seg = acc.n_seg < max_seg ? acc.n_seg : max_seg;
    free(acc.seg); free(acc.tok); free(acc.text); free(text); free(dc.tokens);
    return 0;
}

/* whisper_full_with_state, greedy strategy with best_of = 1 (lib.rs:624): one decoder, argmax at t = 0, std::discrete_distribution
 * draws from the decoder's mt19937 on the fallback passes.  DEVIATION D2': the generator is seeded (0) per call; whisper.cpp seeds it
 * when the state is created and lets it run on across calls. */
static int full_impl(const skwo_model* m, const skwo_params* p, const float* pcm, int n_samples, skwo_result* out, uint32_t* rng_state);
int skwo_full(const skwo_model* m, const skwo_params* p, const float* pcm, int n_samples, skwo_result* out) { return full_impl(m, p, pcm, n_samples, out, NULL); }
/* whisper_full_with_state on a state whose generator has already been used: rng_state = mt[624] + index (625 words; std::mt19937(0) to begin with: whisper_init_state), advanced
 * by this call exactly as decoder 0's generator is */
int skwo_full_rng(const skwo_model* m, const skwo_params* p, const float* pcm, int n_samples, skwo_result* out, uint32_t* rng_state) { return full_impl(m, p, pcm, n_samples, out, rng_state); }
static int full_impl(const skwo_model* m, const skwo_params* p, const float* pcm, int n_samples, skwo_result* out, uint32_t* rng_state) {
    memset(out, 0, sizeof *out); out->min_margin = INFINITY;
#ifdef _OPENMP
    if (p->n_threads > 0) omp_set_num_threads(p->n_threads);
#endif
    int n_len = 0, n_len_org = 0; float* mel = skwo_log_mel(m, pcm, n_samples, &n_len, &n_len_org);
    const int seek_start = 0, seek_end = n_len_org;
    acc_t acc; memset(&acc, 0, sizeof acc);
    const int NV0 = m->hp.n_vocab;
    /* language "auto" (whisper_lang_auto_detect_with_state, offset 0): encode the first window, decode [sot], take the language
     * token with the largest logit (first on a tie); done before the "input is too short" test, as whisper_full_with_state does */
    int lang_id = p->lang_id;
    if (lang_id < 0) {
        if (NV0 < 51865) { free(mel); return -3; }
        if (n_len_org > 0) {
            float* eo = xmalloc_f((size_t)m->hp.n_audio_ctx * m->hp.n_audio_state);
            float* k0 = xmalloc_f((size_t)m->hp.n_text_layer * m->hp.n_audio_ctx * m->hp.n_text_state); float* v0 = xmalloc_f((size_t)m->hp.n_text_layer * m->hp.n_audio_ctx * m->hp.n_text_state);
            float* lg = xmalloc_f(NV0);
            skwo_encode(m, mel, n_len, 0, 0, eo, k0, v0);
            skwo_dec* ds0 = skwo_dec_new(m, k0, v0); int32_t sot = m->tok_sot; skwo_dec_step(ds0, &sot, 1, 0, 0, lg); skwo_dec_free(ds0);
            const int n_lang = N_LANG + (NV0 - 51865 > 0 ? NV0 - 51865 : 0);
            lang_id = 0; for (int i = 1; i < n_lang; ++i) if (lg[m->tok_sot + 1 + i] > lg[m->tok_sot + 1 + lang_id]) lang_id = i;
            free(eo); free(k0); free(v0); free(lg);
        } else lang_id = 0;
    }
    out->lang_id = lang_id;
    if (seek_end < seek_start + DELTA_MIN) { free(mel); return 0; } /* "input is too short" (< 100 ms) */
    const int d = m->hp.n_text_state, nc = m->hp.n_audio_ctx, NV = m->hp.n_vocab;
    float* enc_out = xmalloc_f((size_t)nc * m->hp.n_audio_state);
    float* ck = xmalloc_f((size_t)m->hp.n_text_layer * nc * d); float* cv = xmalloc_f((size_t)m->hp.n_text_layer * nc * d);
    float* raw = xmalloc_f(NV);
    decoder_t dc; memset(&dc, 0, sizeof dc); dc.logits = xmalloc_f(NV); dc.logprobs = xmalloc_f(NV); dc.probs = xmalloc_f(NV); dc.cap = 512;
    dc.tokens = (skwo_token*)malloc(dc.cap * sizeof(skwo_token)); dc.min_margin = INFINITY;
    int32_t prompt_init[8]; int n_prompt = 0;
    prompt_init[n_prompt++] = m->tok_sot;
    if (NV >= 51865) { prompt_init[n_prompt++] = m->tok_sot + 1 + lang_id; prompt_init[n_prompt++] = p->translate ? m->tok_translate : m->tok_transcribe; }
    if (p->no_timestamps) prompt_init[n_prompt++] = m->tok_not;
    int seek = seek_start;
    /* prompt_past: text already produced in this call conditions the next window ([prev] + its last n_text_ctx/2 tokens + the
     * initial prompt) whenever the pass runs at t < 0.5.  whisper_full_default_params sets no_context = true, which only clears
     * it at the start of a call; n_max_text_ctx keeps its default (16384), so the cap is n_text_ctx/2. */
    int32_t* prompt_past = (int32_t*)malloc(sizeof(int32_t) * 1024); int n_past_tok = 0;
    int32_t prompt[512];
    char* text = (char*)malloc(1 << 16);
    mt19937_t rng; mt_seed(&rng, 0);
    if (rng_state) { memcpy(rng.mt, rng_state, sizeof rng.mt); rng.idx = (int)rng_state[624]; }
    while (1) {
        if (seek + DELTA_MIN >= seek_end) break;   /* "if only 100ms left, then stop" */
        skwo_encode(m, mel, n_len, seek, 0, enc_out, ck, cv);
        out->n_windows++;
        /* temperature ladder (whisper_full_with_state): greedy at t = temperature, then sampled passes at +temperature_inc while a pass fails */
        float temps[16]; int n_temps = 0; temps[n_temps++] = p->temperature;
        if (p->temperature_inc > 0.0f) for (float t = p->temperature + p->temperature_inc; t < 1.0f + 1e-6f && n_temps < 16; t += p->temperature_inc) temps[n_temps++] = t;
        float no_speech_prob = 0.0f; int last_take = 0;
        for (int it = 0; it < n_temps; ++it) {
            const float t_cur = temps[it];
            int n_prompt_cur = 0, n_take = 0;
            if (n_past_tok > 0 && t_cur < 0.5f) {
                n_take = m->hp.n_text_ctx / 2 < n_past_tok ? m->hp.n_text_ctx / 2 : n_past_tok;
                { const int room = m->hp.n_text_ctx - (m->hp.n_text_ctx / 2 - 4) - n_prompt - 1;
                if (n_take > room) n_take = room; }   /* only binds with no_timestamps (4-token init): keeps every position inside n_text_ctx */
                prompt[n_prompt_cur++] = m->tok_prev;
                for (int i = 0; i < n_take; ++i) prompt[n_prompt_cur++] = prompt_past[n_past_tok - n_take + i];
            }
            for (int i = 0; i < n_prompt; ++i) prompt[n_prompt_cur++] = prompt_init[i];
            last_take = n_take;
            skwo_dec* ds = skwo_dec_new(m, ck, cv);
            dc.n_tokens = 0; dc.result_len = 0; dc.sum_logprobs_all = 0.0; dc.sum_logprobs = -INFINITY; dc.avg_logprobs = -INFINITY; dc.entropy = 0.0; dc.score = -INFINITY;
            dc.seek_delta = 100 * WHISPER_CHUNK_SIZE; dc.has_ts = 0; dc.failed = 0; dc.completed = 0;
            skwo_dec_step(ds, prompt, n_prompt_cur, 0, 0, raw); out->n_decode_steps++;
            process_logits(m, p, &dc, raw, &no_speech_prob, t_cur);
            const int n_max = m->hp.n_text_ctx / 2 - 4;
            for (int i = 0; i < n_max; ++i) {
                skwo_token tk = (t_cur < 1e-6f) ? sample_best(m, &dc) : sample_dist(m, &dc, &rng);
                if (dc.n_tokens == dc.cap) { dc.cap *= 2; dc.tokens = (skwo_token*)realloc(dc.tokens, dc.cap * sizeof(skwo_token)); }
                dc.tokens[dc.n_tokens++] = tk; dc.sum_logprobs_all += tk.plog;
                if (token_loop_update(m, p, &dc, tk.id, i, seek, seek_end, n_max)) break;
                { int32_t t = tk.id; skwo_dec_step(ds, &t, 1, n_prompt_cur + i, 0, raw); out->n_decode_steps++; }
                process_logits(m, p, &dc, raw, &no_speech_prob, t_cur);
            }
            skwo_dec_free(ds);
            if (!dc.failed) {
                dc.n_tokens = dc.result_len; sequence_score(&dc);
                if (dc.result_len > 32 && dc.entropy < p->entropy_thold) dc.failed = 1;
            }
            if (dc.failed || (dc.avg_logprobs < p->logprob_thold && no_speech_prob < p->no_speech_thold)) out->fallback_requested++;   /* this pass failed */
            else break;
        }
        /* output */
        {
            int seek_delta = dc.seek_delta; const int result_len = dc.result_len; const skwo_token* tc = dc.tokens; const int ntc = dc.n_tokens; (void)result_len;
            const int is_no_speech = (no_speech_prob > p->no_speech_thold && dc.avg_logprobs < p->logprob_thold);
            { /* update prompt_past (recalled): clear; re-insert what this window's prompt took from it (prompt.begin()+1 .. end()-prompt_init.size())
               * unconditionally; append this window's tokens only `&& !is_no_speech` */
                int32_t keep[512]; for (int i = 0; i < last_take; ++i) keep[i] = prompt_past[n_past_tok - last_take + i];
                n_past_tok = 0;
                for (int i = 0; i < last_take; ++i) prompt_past[n_past_tok++] = keep[i];
                if (!is_no_speech) for (int i = 0; i < dc.result_len; ++i) prompt_past[n_past_tok++] = tc[i].id;
            }
            (void)tc; (void)ntc;
            seek_delta = window_output(m, p, &dc, seek, seek_end, is_no_speech, &acc, text);
            seek += seek_delta;
        }
        if (dc.min_margin < out->min_margin) out->min_margin = dc.min_margin;
    }
    free(prompt_past); free(text); free(mel); free(enc_out); free(ck); free(cv); free(raw); free(dc.logits); free(dc.logprobs); free(dc.probs); free(dc.tokens);
    out->n_segments = acc.n_seg; out->segments = acc.seg; out->n_tokens = acc.n_tok; out->tokens = acc.tok; out->text = acc.text; out->text_len = acc.n_text;
    if (rng_state) { memcpy(rng_state, rng.mt, sizeof rng.mt); rng_state[624] = (uint32_t)rng.idx; }
    return 0;
}
void skwo_result_free(skwo_result* r) { free(r->segments); free(r->tokens); free(r->text); memset(r, 0, sizeof *r); }

/* ------------------------------------------------ R1: rubato FastFixedIn */
#define POLY_LEN 8
struct skwo_resampler { int ch, chunk; double last_index, ratio; float* buf; /* [ch][chunk + 2*POLY_LEN] */ };
skwo_resampler* skwo_resampler_new(double ratio, int chunk_frames, int channels) {
    skwo_resampler* r = (skwo_resampler*)calloc(1, sizeof *r); r->ch = channels; r->chunk = chunk_frames; r->ratio = ratio;
    r->last_index = -(double)(POLY_LEN / 2); r->buf = (float*)calloc((size_t)channels * (chunk_frames + 2 * POLY_LEN), sizeof(float)); return r;
}
void skwo_resampler_free(skwo_resampler* r) { if (r) { free(r->buf); free(r); } }
int skwo_resampler_process(skwo_resampler* r, const float* in_planar, float* out_planar, int out_cap) {
    const int cs = r->chunk, bl = cs + 2 * POLY_LEN;
    for (int c = 0; c < r->ch; ++c) { float* b = r->buf + (size_t)c * bl; memmove(b, b + cs, sizeof(float) * 2 * POLY_LEN); memcpy(b + 2 * POLY_LEN, in_planar + (size_t)c * cs, sizeof(float) * cs); }
    double idx = r->last_index; const double t_ratio = 1.0 / r->ratio; /* fixed ratio: t_ratio_increment = 0 */
    const double end_idx = (double)(cs - (POLY_LEN + 1)) - ceil(t_ratio);
    int n = 0;
    while (idx < end_idx) {
        idx += t_ratio;
        const double fl = floor(idx); const long start = (long)fl; const float frac = (float)(idx - fl);
        if (n >= out_cap) return -1;
        for (int c = 0; c < r->ch; ++c) { const float* b = r->buf + (size_t)c * bl + (start + 2 * POLY_LEN); out_planar[(size_t)c * out_cap + n] = (1.0f - frac) * b[0] + frac * b[1]; }
        n++;
    }
    r->last_index = idx - (double)cs;
    return n;
}

/* arithmetic-contract probes: the CPU side of tests/test_gpu_math.py */
void skwo_math(const skwo_model* m, int kind, const float* in, float* out, long n) {
    for (long i = 0; i < n; ++i) {
        float x = in[i], y;
        if (kind == 0) y = skw_expf(x);
        else if (kind == 1) y = skw_logf(x);
        else if (kind == 2 || kind == 3) y = skw_round_f16(x);
        else if (kind == 4) y = skw_gelu_lookup(x, m->gelu_tab);
        else if (kind == 5) y = 1.0f / sqrtf(x + 1e-5f);
        else if (kind == 6) y = (float)(1.0 / (double)x);
        else y = (float)log10((double)x);
        out[i] = y;
    }
}

/* ---- the f16 matrix cores' arithmetic, restated (include/skw_mfma_model.h): checker for the f16_mfma precision's contractions ---- */
#include "../include/skw_mfma_model.h"
float skwo_mfma_f16_element(const uint16_t* a32, const uint16_t* b32, float c) { return skw_mfma_f32_16x16x32_f16_element(a32, b32, c); }
void skwo_mfma_f16_elements(const uint16_t* a, const uint16_t* b, const float* c, float* d, long n) {
    for (long i = 0; i < n; ++i) d[i] = skw_mfma_f32_16x16x32_f16_element(a + 32 * i, b + 32 * i, c[i]);
}
void skwo_mfma_f16_tiles(const uint16_t* A, const uint16_t* B, const float* C, float* D, long P) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                uint16_t b[32];
                for (int k = 0; k < 32; ++k) b[k] = B[(p * 32 + k) * 16 + j];
                D[(p * 16 + i) * 16 + j] = skw_mfma_f32_16x16x32_f16_element(A + (p * 16 + i) * 32, b, C[(p * 16 + i) * 16 + j]);
            }
}
int skwo_gemm_f16mfma(const uint16_t* A, long lda, const uint16_t* W, long ldw, int M, int N, int K, int n_split, float* C, long ldc) {
    if (M < 1 || N < 1 || n_split < 1 || K % (32 * n_split)) return -1;
    const int kpart = K / n_split;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float tot = 0.0f;
            for (int s = 0; s < n_split; ++s) {
                float acc = 0.0f;
                for (int k0 = s * kpart; k0 < (s + 1) * kpart; k0 += 32) acc = skw_mfma_f32_16x16x32_f16_element(A + (long)m * lda + k0, W + (long)n * ldw + k0, acc);
                tot = s ? tot + acc : acc;
            }
            C[(long)m * ldc + n] = tot;
        }
    return 0;
}
