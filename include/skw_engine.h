/*
 * skw_engine.h — C ABI of the MI355X Whisper engine (libskw_engine.so).
 *
 * This is the drop-in boundary for the arithmetic the reference reaches through
 * whisper-rs (FFI over whisper.cpp's C API).  Each entry point names the
 * reference call it replaces; a Rust host binds these with a ~40-line
 * `extern "C"` block (INTEGRATION.md shows it next to the whisper-rs calls it
 * displaces).  Plain pointers and sizes only — no torch, no C++ types.
 *
 *   reference (plugins/native/whisper/src/lib.rs)            this library
 *   ---------------------------------------------            ----------------------------
 *   WhisperContext::new_with_params(path, params)  :354-363  skw_model_load(path, gpu_device)
 *   context.create_state()                         :377-379  skw_ctx_create(model, max_batch, ...)
 *   FullParams::new(Greedy{best_of:1}) + setters   :624-641  skw_full_params (skw_full_default_params)
 *   whisper_state.full(params, &samples)           :644-646  skw_full_batch(ctx, params, pcm[], n[], n_clips)
 *   state.as_iter() / segment.to_str() /
 *     start_timestamp() / end_timestamp()          :650-660  skw_result.segments[i].{text,t0,t1}
 *
 * skw_full_batch is the batched form of `full`: n_clips independent calls of
 * whisper_full_with_state (greedy strategy, best_of 1, temperature fallback ladder) executed together on one GPU.
 * The kernels fail loudly (non-zero return + message) when no gfx950 device is
 * present; there is no CPU fallback in this library.
 */
#ifndef SKW_ENGINE_H
#define SKW_ENGINE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct skw_model skw_model;
typedef struct skw_ctx skw_ctx;

typedef struct {
    int32_t n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
} skw_hparams;

/* the whisper_full_params fields the reference sets (lib.rs:624-641) + the whisper.cpp defaults that shape greedy decoding */
typedef struct {
    int32_t lang_id;          /* whisper_lang_id(language); "en" = 0; < 0 = whisper.cpp's "auto": detected per clip from the first window (multilingual models) */
    int32_t translate;
    int32_t suppress_blank;
    int32_t suppress_nst;
    int32_t no_timestamps;
    int32_t single_segment;
    int32_t max_tokens;
    float   max_initial_ts;
    float   entropy_thold;
    float   logprob_thold;
    float   no_speech_thold;
    int32_t n_threads;        /* accepted for ABI parity with the reference param; unused on the GPU */
    float   temperature;      /* whisper_full_params.temperature, default 0.0: the first pass of a window is greedy argmax */
    float   temperature_inc;  /* default 0.2: a window that fails (entropy / logprob / repetition rules) is decoded again at +0.2 .. 1.0,
                                 sampling with std::discrete_distribution semantics from a per-clip std::mt19937(0); <= 0 disables the ladder */
} skw_full_params;

typedef struct {
    int64_t t0, t1;             /* centiseconds: whisper_full_get_segment_t0/t1 */
    int32_t tok_begin, tok_end; /* [begin,end) into skw_result.tokens */
    int32_t text_off, text_len; /* into skw_result.text */
} skw_segment;

typedef struct { int32_t id, tid; float p, plog, pt, ptsum;
                 float margin;   /* top1 - top2 admissible logit when this token was chosen (+inf on sampled passes): distance from a tie (diagnostic) */
} skw_token;

typedef struct {
    int32_t n_segments, n_tokens, n_windows, n_decode_steps;
    int32_t fallback_requested; /* decoding passes that failed whisper.cpp's acceptance rules (each but the last temperature's is retried) */
    float   min_margin;         /* smallest top1-top2 admissible-logit margin seen (diagnostic) */
    skw_segment* segments;
    skw_token* tokens;
    char* text;                 /* concatenated segment texts, NUL terminated */
    int32_t text_len;
    int32_t lang_id;            /* the language decoded with: params.lang_id, or the detected one when that was < 0 */
} skw_result;

/* ---- lifetime ---- */
int  skw_device_count(void);
/* page-locked host memory for PCM handed to skw_full_batch: the engine's H2D copies of such buffers are asynchronous DMA (64 x 30 s clips: ~2.3 ms) instead of the driver's
 * staged copy out of pageable memory (~6 ms).  NULL when the allocation fails (callers fall back to ordinary memory). */
void* skw_host_alloc(size_t bytes);
void skw_host_free(void* p);
skw_model* skw_model_load(const char* ggml_path, int device, char* err, size_t errlen);
/* Block-quantised files (q4_0 / q4_1 / q5_0 / q5_1 / q8_0; the reference's default model is a q5_1 file, lib.rs:114-116):
 *   SKW_QUANT_GGML (default)  a context in the exact precision multiplies the way ggml does — activation rows to q8_0 / q8_1 blocks,
 *                             integer block dots, f32 scales (include/skw_ggml_quant.h (a)); a context in the f16_mfma precision runs the
 *                             file's dequantised f16 twin (both representations are resident)
 *   SKW_QUANT_F16_TWIN        the twin in both precisions (round-1 behaviour; also forced by the environment variable SKW_QUANT_TWIN)
 * skw_model_quant_type: the ggml tensor type running in ggml's arithmetic (2, 3, 6, 7, 8), or 0. */
enum { SKW_QUANT_F16_TWIN = 0, SKW_QUANT_GGML = 1 };
skw_model* skw_model_load_ex(const char* ggml_path, int device, int quant_mode, char* err, size_t errlen);
int skw_model_quant_type(const skw_model* m);
void skw_model_free(skw_model*);
void skw_model_get_hparams(const skw_model*, skw_hparams* out);
const char* skw_model_token_text(const skw_model*, int id, int* len);
int  skw_model_lang_id(const char* lang); /* whisper_lang_id; -1 if unknown */

skw_ctx* skw_ctx_create(skw_model*, int max_batch, int max_samples_per_clip, char* err, size_t errlen);
void skw_ctx_free(skw_ctx*);
const char* skw_ctx_last_error(const skw_ctx*);

/* Which form of the dense contractions (K2, K4-K6 and, in decode, K8-K10) a context runs.  Additive: the reference has no such knob.
 *   SKW_PRECISION_EXACT     f16 values widened to f32 and chained in k order on the f32-input MFMA: every tensor is bit-identical
 *                           to oracle/ (the checker mode; 1/16 of the f16 matrix rate);
 *   SKW_PRECISION_F16_MFMA  the same f16 operands fed to v_mfma_f32_16x16x32_f16 (f32 accumulate, hardware summation order):
 *                           intermediate tensors agree with the exact mode's to the tolerances tests/test_gpu_f16.py states, and every
 *                           greedy decision equals the exact mode's given the same history unless the exact mode's own top1 - top2
 *                           logit margin at that step is below twice the measured logit error (checked step by step under teacher
 *                           forcing, skw_full_batch_traced).  A free-running transcript is therefore identical to the exact mode's
 *                           up to its first such near-tie and may differ after it (on the synthetic benchmark model about a quarter
 *                           of the 30 s clips contain one; bench.py reports the count).  log-mel and the logit rules are the exact
 *                           kernels in both modes, and so are the encoder's LayerNorms; GELU is evaluated (exp2 / rcp) instead of looked
 *                           up, and the DECODE step folds its LayerNorms into the consuming GEMMs (k_gemm16_small_lnA: one-pass
 *                           E[x^2] - mean^2 statistics in f64, f32 rstd — within the mode's tolerance of k_layernorm's two-pass
 *                           arithmetic, not identical to it; SKW_DEC_LN_STATS=0 restores the LayerNorm kernels). */
#define SKW_PRECISION_EXACT 0
#define SKW_PRECISION_F16_MFMA 1
int skw_ctx_set_precision(skw_ctx*, int precision);   /* 0 on success; takes effect from the next call on this context */
int skw_ctx_get_precision(const skw_ctx*);

void skw_full_default_params(skw_full_params*);

/* ---- the hot path ---- */
/* pcm[i]: 16 kHz mono f32, n_samples[i] samples; host pointers (pcm_on_device = 0) or device pointers (= 1).
 * results[i] is filled and must be released with skw_result_free. n_clips <= max_batch. Returns 0 on success. */
int  skw_full_batch(skw_ctx*, const skw_full_params*, const float* const* pcm, const int32_t* n_samples, int n_clips,
                    int pcm_on_device, skw_result* results);
void skw_result_free(skw_result*);
/* The temperature ladder's generator as the reference keeps it.  whisper.cpp owns ONE std::mt19937 per whisper_state (decoder 0's, seeded with 0 by whisper_init_state) and lets
 * it run on across calls of whisper_full_with_state; the reference node creates one state per instance (plugins/native/whisper/src/lib.rs:377-379), so an instance's n-th segment
 * that needs a sampled pass draws from where its earlier segments left the stream.  skw_full_batch_rng is skw_full_batch with that stream handed in and out per clip:
 * rng_state[i] = NULL (seed 0 for this call, what skw_full_batch does) or SKW_RNG_STATE_WORDS words — mt[624] and the index, initialised once by skw_rng_state_init — which the
 * call advances exactly as the reference's generator would advance on this clip.  Rows of a batch never share a stream, so the result does not depend on batch composition. */
#define SKW_RNG_STATE_WORDS 625
void skw_rng_state_init(uint32_t* state /* [SKW_RNG_STATE_WORDS] */);
int  skw_full_batch_rng(skw_ctx*, const skw_full_params*, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device,
                        uint32_t* const* rng_state /* [n_clips], entries may be NULL */, skw_result* results);

/* ---- decision trace / teacher forcing (parity instrumentation of the hot path; the reference has no counterpart) ----
 * skw_full_batch_traced is skw_full_batch that also returns, per clip, one record for EVERY sampling decision it made, in execution order
 * over all windows and temperature passes (discarded tokens included).  With forced_ids[i] == NULL the run is free (forced_id == chosen_id).
 * With forced_ids[i] = the chosen_id sequence of another run (another precision of the same model on the same audio), the decoder is fed
 * those tokens instead of its own choices, so every step sees exactly the history the other run saw and the two runs' decisions can be
 * compared step by step: chosen_id is what THIS precision's argmax picked, top1 / top2 the two largest admissible logits, forced_logit
 * the logit of the fed token.  A forced sequence that runs out before the decoder stops is an error (the runs' control flow diverged).
 * The decode steps are launched eagerly in this mode (same kernels as the captured step graph, plus the trace form of the sampler). */
typedef struct { int32_t chosen_id, forced_id, top1_id, top2_id; float top1, top2, forced_logit, lse;
                 float temperature;   /* of the pass this decision belongs to; > 0: chosen_id is a std::discrete_distribution draw (not an argmax) and the logits here are the row's divided by it */
                 int32_t pad; } skw_trace_step;
typedef struct { int32_t n; skw_trace_step* steps; } skw_trace;
int  skw_full_batch_traced(skw_ctx*, const skw_full_params*, const float* const* pcm, const int32_t* n_samples, int n_clips, int pcm_on_device,
                           const int32_t* const* forced_ids /* [n_clips] or NULL */, const int32_t* n_forced /* [n_clips] or NULL */,
                           skw_trace* traces /* [n_clips] out; release with skw_trace_free */, skw_result* results);
void skw_trace_free(skw_trace*);

/* timing of the last skw_full_batch (milliseconds, GPU events on the engine's stream) */
typedef struct { float mel_ms, encode_ms, decode_ms, total_ms; int32_t n_windows, n_decode_steps, n_tokens;
                 /* n_row_steps: sum over rows of the decode steps the row was live in (a finished row's attention kernels return at once:
                    the HBM bytes of a step scale with its live rows) */
                 int32_t n_row_steps;
                 /* how the decode step of the call's last window pass ran: row groups (each a step graph on its own stream) and rows in the first group — what a per-launch
                    roofline of the decode kernels has to be booked against (bench.py reads these instead of restating the engine's default) */
                 int32_t decode_groups, decode_group_rows;
} skw_timing;
void skw_ctx_last_timing(const skw_ctx*, skw_timing* out);
/* per-kernel-class event timing (adds an event pair around every launch; use for roofline accounting, not for the timed run).
 * classes: 0 k_gemm, 1 k_gemm_smallm, 2 k_attn_encoder, 3 k_layernorm, 4 k_mel, 5 k_dec_self_attn, 6 k_dec_sample, 7 other, 8 k_dec_cross_attn */
void skw_ctx_profile(skw_ctx*, int on);
int  skw_ctx_profile_get(skw_ctx*, int cls, char* name, size_t name_len, long* count, double* ms, double* algorithmic_flops, double* algorithmic_bytes);
/* In-kernel launch clock of the decode step's dominant kernel (the f16_mfma cross attention).  HIP events cannot sit between the kernels of a captured step graph, and an eager,
 * event-stamped step is not the timed configuration (two row groups interleave differently): armed, the kernel itself records first-wave-in and last-wave-out of every launch on the
 * device's constant-rate clock while the step graphs run as in any other call.  skw_ctx_kernel_clock_get: the launches the last skw_full_batch recorded — count, summed microseconds,
 * summed live rows (a launch's algorithmic bytes = 4 B x live rows x n_audio_ctx x n_text_state), shortest / longest launch, the clock's rate in kHz.  Diagnostic; off by default. */
int  skw_ctx_kernel_clock(skw_ctx*, int on);
int  skw_ctx_kernel_clock_get(skw_ctx*, long* launches, double* sum_us, double* sum_live_rows, double* min_us, double* max_us, int* clock_khz);
/* every recorded launch as (begin us, end us, live rows), relative to the earliest begin; returns the count written.  Row groups run on their own streams, so launches overlap:
 * the union of the intervals is the time the kernel was in flight at all. */
long skw_ctx_kernel_clock_records(skw_ctx*, double* out /* [cap][3] */, long cap);
/* the HIP stream the engine launches on (opaque hipStream_t) */
void* skw_ctx_stream(const skw_ctx*);

/* ---- stage taps (used by the parity tests and by hosts that only need a front end) ---- */
/* K1: log-mel of one clip; mel_out [n_mel][n_len] f32 (whisper.cpp layout); caller provides cap floats */
int skw_log_mel(skw_ctx*, const float* pcm_host, int n_samples, float* mel_out, size_t cap, int* n_len, int* n_len_org);
/* K2: conv stem + positional embedding for the window at `seek`; x0 [n_audio_ctx][n_state] f32 */
int skw_conv_stem(skw_ctx*, const float* pcm_host, int n_samples, int seek, float* x0);
/* K2-K6: encoder output (after ln_post) [n_audio_ctx][n_state] f32 and, when non-null, cross K/V [n_text_layer][n_audio_ctx][n_state] f32 */
int skw_encode(skw_ctx*, const float* pcm_host, int n_samples, int seek, float* enc_out, float* cross_k, float* cross_v);
/* K7-K10: after skw_encode, run the decoder on tokens[0..n) from position 0 and return the last token's logits [n_vocab] */
int skw_decode_logits(skw_ctx*, const int32_t* tokens, int n_tokens, float* logits);

/* ---- resampler front end: the arithmetic of the reference's audio::resampler node on the GPU ----
 * reference: crates/nodes/src/audio/filters/resampler.rs:231-244 (FastFixedIn::new(ratio, 1.0, Linear, chunk_frames, channels)),
 *            :384-514 (per-chunk process), :543-688 (remainder with a fresh resampler). */
typedef struct skw_dsp skw_dsp;                 /* model-free device context (stream + scratch) */
skw_dsp* skw_dsp_create(int device, char* err, size_t errlen);
void skw_dsp_free(skw_dsp*);
const char* skw_dsp_last_error(const skw_dsp*);
typedef struct { double last_index; double ratio; int32_t chunk_frames; int32_t channels; float hist[32]; } skw_resampler_state; /* rubato's carried state: fractional index + 16-frame history */
void skw_resampler_init(skw_resampler_state*, double ratio /* out/in */, int chunk_frames, int channels);
/* n_chunks full chunks of interleaved f32 input -> interleaved output (host pointers); bit-exact with rubato's Linear interpolation */
int skw_resample_linear(skw_dsp*, skw_resampler_state*, const float* in, int n_chunks, float* out, int out_cap_frames, int* out_frames);
/* how the last skw_resample_linear obtained its index sequence: 0 = the closed-form per-chunk proposal was proven equal to rubato's
 * sequential f64 walk on the device (48 / 32 / 96 kHz sources), 1 = a second proposal was (44.1 kHz family and every other ratio met so
 * far: chunk starts walked by the host for long calls, stepped binade by binade on the device for short ones), 2 = neither: the
 * single-lane walk ran */
int skw_dsp_last_scan_fallback(const skw_dsp*);
/* additive quality mode: polyphase Kaiser-windowed sinc (32 taps/phase), whole buffer */
int skw_resample_polyphase(skw_dsp*, const float* in, long n_in_frames, int channels, int in_rate, int out_rate, float* out, long out_cap_frames, long* out_frames);
/* the same filter over a stream, packet by packet: the input tail later outputs need stays in HBM; results identical to the whole-buffer call */
typedef struct skw_pp_stream skw_pp_stream;
skw_pp_stream* skw_polyphase_stream_create(skw_dsp*, int channels, int in_rate, int out_rate);
int  skw_polyphase_stream_push(skw_pp_stream*, const float* in, long n_frames, int final_call, float* out, long out_cap_frames, long* out_frames);
void skw_polyphase_stream_free(skw_pp_stream*);

#ifdef __cplusplus
}
#endif
#endif
