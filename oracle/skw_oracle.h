/*
 * oracle/skw_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic the reference's Whisper node executes.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; nothing under streamkit_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED: the reference keeps this arithmetic in un-vendored
 * third-party code (whisper.cpp via whisper-rs 0.15.1 / whisper-rs-sys 0.14.1,
 * /root/reference/plugins/native/whisper/Cargo.lock:1174-1185; rubato 0.16.2,
 * /root/reference/Cargo.lock:3708-3711), ships no model weights, and none of its
 * tests pins a value on this path.  What is restated here is
 * whisper.cpp's published algorithm (log_mel_spectrogram, whisper_encode_internal,
 * whisper_decode_internal, whisper_process_logits, whisper_full_with_state) and
 * rubato's FastFixedIn/Linear, anchored on the reference's own call sites:
 *   plugins/native/whisper/src/lib.rs:404-494  (VAD framing + segment cuts)
 *   plugins/native/whisper/src/lib.rs:582-702  (FullParams + full() + segments)
 *   crates/nodes/src/audio/filters/resampler.rs:231-244, 384-514, 543-688
 * and on the behavioural goldens derivable from those files alone
 * (tests/golden/, SURVEY.md §8c last row).
 */
#ifndef SKW_ORACLE_H
#define SKW_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct skwo_model skwo_model;

typedef struct {
    int32_t n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
} skwo_hparams;

/* whisper_full_params subset that the reference sets (lib.rs:624-641) plus the
 * whisper.cpp defaults that shape greedy decoding */
typedef struct {
    int32_t lang_id;            /* index into whisper's language table; "en" = 0; < 0 = "auto" (whisper_lang_auto_detect_with_state) */
    int32_t translate;          /* lib.rs:626 -> false */
    int32_t suppress_blank;     /* lib.rs:633 */
    int32_t suppress_nst;       /* lib.rs:634 */
    int32_t no_timestamps;      /* whisper.cpp default false */
    int32_t single_segment;     /* default false */
    int32_t max_tokens;         /* default 0 */
    float   max_initial_ts;     /* default 1.0 */
    float   entropy_thold;      /* default 2.4 */
    float   logprob_thold;      /* default -1.0 */
    float   no_speech_thold;    /* default 0.6 */
    int32_t n_threads;
    float   temperature;        /* default 0.0: first pass is greedy argmax */
    float   temperature_inc;    /* default 0.2: fallback ladder 0.2, 0.4 .. 1.0 with sampling; <= 0 disables the ladder */
} skwo_params;

typedef struct {
    int64_t t0, t1;             /* centiseconds, as whisper_full_get_segment_t0/t1 */
    int32_t tok_begin, tok_end; /* [begin,end) into the token array */
    int32_t text_off, text_len; /* into the text buffer */
} skwo_segment;

typedef struct {
    int32_t id, tid;
    float p, plog, pt, ptsum;
    float margin;                /* top1 - top2 admissible logit at this step (+inf on sampled passes): how close the argmax was to a tie */
} skwo_token;

typedef struct {
    int32_t n_segments, n_tokens, n_windows, n_decode_steps;
    int32_t fallback_requested;  /* decoding passes that failed the acceptance rules (all but the last temperature's are retried) */
    float   min_margin;          /* smallest top1-top2 logit margin over all sampled steps */
    skwo_segment* segments;
    skwo_token* tokens;          /* all result tokens in order (timestamps included) */
    char* text;                  /* concatenated segment texts (no separators) */
    int32_t text_len;
    int32_t lang_id;             /* language decoded with (detected when params.lang_id < 0) */
} skwo_result;

/* how a uniformly block-quantised file (q4_0 / q4_1 / q5_0 / q5_1 / q8_0 matmul weights) is multiplied; set BEFORE skwo_load:
 *   1 (default) ggml's arithmetic: activations to q8 blocks, integer block dots (include/skw_ggml_quant.h (a));  0: the dequantised f16 twin */
void skwo_set_quant_mode(int mode);
skwo_model* skwo_load(const char* path, char* err, int errlen);
int skwo_model_quant(const skwo_model* m);
int skwo_debug_linear_q8(int type, const uint8_t* blocks, int n_out, int n_in, const float* A, int rows, float* out);
/* test hook: one quantised mul_mat */   /* ggml type running in q8 arithmetic, 0 if none */
void skwo_free(skwo_model*);
void skwo_get_hparams(const skwo_model*, skwo_hparams* out);
const char* skwo_token_str(const skwo_model*, int id, int* len);
void skwo_default_params(skwo_params* p);

/* K1: whisper.cpp log_mel_spectrogram. Returns malloc'd [n_mel][n_len]; caller frees with skwo_free_buf */
float* skwo_log_mel(const skwo_model*, const float* pcm, int n_samples, int* n_len, int* n_len_org);
void skwo_free_buf(void*);

/* K2-K6: encoder over mel frames [seek, seek+3000). enc_out: [n_audio_ctx][n_state] f32 (after ln_post).
 * cross_k/cross_v (optional, may be NULL): [n_text_layer][n_audio_ctx][n_state] f32 holding f16-rounded values */
int skwo_encode(const skwo_model*, const float* mel, int n_len, int seek, int n_threads,
                float* enc_out, float* cross_k, float* cross_v);

/* debug taps for kernel-level parity tests */
int skwo_conv_stem(const skwo_model*, const float* mel, int n_len, int seek, int n_threads, float* x0 /*[n_ctx][n_state]*/);

/* K7-K10: run the decoder on `n_tokens` tokens appended at position n_past for ONE sequence whose
 * cross K/V were produced by skwo_encode; returns logits of the last token. State is opaque. */
typedef struct skwo_dec skwo_dec;
skwo_dec* skwo_dec_new(const skwo_model*, const float* cross_k, const float* cross_v);
void skwo_dec_free(skwo_dec*);
void skwo_dec_reset(skwo_dec*);
int skwo_dec_step(skwo_dec*, const int32_t* tokens, int n_tokens, int n_past, int n_threads, float* logits /*[n_vocab]*/);

/* K11-K12 + W4: whisper_full_with_state, greedy best_of=1, T=0 pass */
int skwo_full(const skwo_model*, const skwo_params*, const float* pcm, int n_samples, skwo_result* out);
/* the same on a state whose std::mt19937 has run before: rng_state[625] = mt[624] + index, in and out (NULL: seeded with 0 for this call) */
int skwo_full_rng(const skwo_model*, const skwo_params*, const float* pcm, int n_samples, skwo_result* out, uint32_t* rng_state);
/* test hook: n_draws of std::discrete_distribution<>(probs, probs + n) from std::mt19937(seed), as restated in skw_oracle.c */
int skwo_discrete_draw(const float* probs, int n, uint32_t seed, int n_draws, int32_t* out);
void skwo_result_free(skwo_result*);
/* test hooks for the K11 rules on their own (tests/test_cpu_logit_rules.py, tests/golden/make_logit_rule_goldens.py): one call of
 * whisper_process_logits + whisper_sample_token(best) on caller-supplied logits for the decoder history hist[0..n_hist); flags bit 0
 * leaves the timestamp-mass rule out; -1 if the token loop could not have produced the history */
int skwo_debug_process_logits(const skwo_model*, const skwo_params*, const int32_t* hist, int n_hist, const float* raw_logits, float temperature, int flags,
                              float* out_logits, float* out_logprobs, skwo_token* chosen, float* no_speech_prob);
/* kind 0: always-suppressed specials; 1: the suppress_nst list; 2: the blank rule's pair; 3: {eot, notimestamps, timestamp_begin, nospeech, sot} */
int skwo_debug_rule_ids(const skwo_model*, int kind, int32_t* ids, int cap);

/* teacher-forced logits (for margin diagnostics): runs window at `seek` with given token prefix */

/* debug taps of encoder layer 0 (l0.ln1, l0.q, l0.k, l0.v, l0.att, l0.x1, l0.ln2, l0.h, l0.x2), natural layouts */
/* C[m][n] = the f16 matrix cores' contraction of A[m][:] and W[n][:] as the f16_mfma GEMM kernels order it (include/skw_mfma_model.h): operands in MEMORY order (slot 8 g + e of
 * a 32-block = memory position 8 g + e: what a lane group loads), one v_mfma_f32_16x16x32_f16 per 32-block in ascending order from a zero accumulator; n_split > 1: the K axis
 * in n_split contiguous parts, each chained from zero, the partial sums added in f32 in ascending order — the decode kernels' four waves.  K % (32 n_split) == 0. */
int skwo_gemm_f16mfma(const uint16_t* A, long lda, const uint16_t* W, long ldw, int M, int N, int K, int n_split, float* C, long ldc);
/* one instruction's output element: 32 operand pairs in slot order + accumulator (tests pin this against committed hardware vectors) */
float skwo_mfma_f16_element(const uint16_t* a32, const uint16_t* b32, float c);
/* P whole instructions: A P x [16][32] (row i, slot k), B P x [32][16] (slot k, column j), C / D P x [16][16] */
/* n single elements: a / b n x [32] in slot order, c / d n */
void skwo_mfma_f16_elements(const uint16_t* a, const uint16_t* b, const float* c, float* d, long n);
void skwo_mfma_f16_tiles(const uint16_t* A, const uint16_t* B, const float* C, float* D, long P);
/* test hook: one window's token-loop bookkeeping and segment assembly on a given stream of sampled token ids (skw_oracle.c) */
int skwo_debug_window(const skwo_model* m, const skwo_params* p, const int32_t* toks, int n, int seek, int seek_end, int64_t* seg_t, int32_t* seg_off, int32_t* seg_tokens,
                      int max_seg, int* n_seg, int* advance, int* n_kept, int* n_consumed, int* failed);
void skwo_debug_enable(int on);
long skwo_debug_get(const char* name, float* out, size_t cap);

/* W1-W3: segmentation state machine over per-frame speech probabilities; cuts[i] = {start_ms, end_ms, n_samples, reason (0 max_duration, 1 silence), silence_ms or -1, frame index} */
int skwo_segment_sim(const float* prob, int n_frames, float threshold, uint64_t min_silence_ms, float max_secs, int64_t* cuts, int max_cuts);

/* R1-R3: rubato FastFixedIn<f32>, PolynomialDegree::Linear, as driven by resampler.rs */
typedef struct skwo_resampler skwo_resampler;
skwo_resampler* skwo_resampler_new(double ratio, int chunk_frames, int channels);
void skwo_resampler_free(skwo_resampler*);
/* planar in [ch][chunk_frames] -> planar out [ch][cap]; returns frames written per channel */
int skwo_resampler_process(skwo_resampler*, const float* in_planar, float* out_planar, int out_cap);

#ifdef __cplusplus
}
#endif
#endif
