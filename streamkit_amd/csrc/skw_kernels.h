// skw_kernels.h — launchers for the hand-written gfx950 kernels of the Whisper hot path.
//
// Arithmetic contract: include/skw_math.h.  All dense contractions run on
// v_mfma_f32_16x16x4_f32 (f16-valued operands widened to f32, f32 accumulate),
// which is bit-for-bit a k-ascending fmaf chain, so results do not depend on
// tiling.  Every f16 GEMM operand keeps its contraction axis in "kperm" order
// (see skw_kperm) so that one 16-byte load hands a lane the eight k values its
// MFMA slot consumes in the next eight instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;

// position (in memory) of logical contraction index k: within each aligned block of 32,
// the 8 values with k % 4 == q are contiguous at [8q, 8q+8).
__host__ __device__ __forceinline__ int skw_kperm(int k) { return (k & ~31) | ((k & 3) << 3) | ((k >> 2) & 7); }
// Fragment-order cross K / V^T (f16_mfma precision).  The decode step's cross attention streams 295 MB per launch; read as rows, one wave-instruction fetches 16 rows x 64 B
// (an MFMA operand tile) — twice the requests per byte of a full-line stream.  Stored in the order the matrix core takes them, every load instruction is one contiguous KiB:
//   K    per (slot, head): Tpad / 16 key tiles x 2 KiB = [d half kk][lane i + 16 g][8 halves]: lane row i holds key 16 T + 4 (i & 3) + (i >> 2) (so that the score MFMA's
//        accumulators come out in V^T's key order), the halves are d = 32 kk + 8 g .. + 7 of that head, natural order (as the query)
//   V^T  per (slot, head): Tpad / 32 key blocks x 4 KiB = [channel tile ct][lane i + 16 g][8 halves]: channel 16 ct + i, memory positions 8 g .. 8 g + 7 of the block (kperm order, as before)
// Both are permutations of the 16-byte chunks of the row layouts.  Offsets in halves; feat / pos multiples of 8.
__host__ __device__ __forceinline__ long skw_kfrag_off(int slot, int H, int Tpad, int key, int feat) {
    const int h = feat >> 6, c = (feat >> 3) & 7, r = key & 15, i = 4 * (r & 3) + (r >> 2);
    return (((long)slot * H + h) * (Tpad >> 4) + (key >> 4)) * 1024 + (c >> 2) * 512 + (i + 16 * (c & 3)) * 8;
}
// fragment-order activation image (SkwGemmArgs::c_frag / a_frag): element offset of position p (multiple of 4) of row m, K halves per row
__host__ __device__ __forceinline__ long skw_afrag_off(int m, int p, int K) { return ((long)(m >> 4) * (K >> 5) + (p >> 5)) * 512 + ((m & 15) + 16 * ((p >> 3) & 3)) * 8 + (p & 7); }
__host__ __device__ __forceinline__ long skw_vtfrag_off(int slot, int H, int Tpad, int feat, int pos) {
    const int h = feat >> 6, ch = feat & 63;
    return (((long)slot * H + h) * (Tpad >> 5) + (pos >> 5)) * 2048 + (ch >> 4) * 512 + ((ch & 15) + 16 * ((pos >> 3) & 3)) * 8;
}

// ---- the switchboard: every alternative path a run can select is one row of ONE table (skw_engine.hip, g_sw_defs).  A switch is read from the environment (SKW_<NAME>) the first
// time anything asks — the only getenv of this library — and tests flip it in-process (skw_debug_switch_set); tests/test_gpu_switches.py runs a parity case per row.
// What the measured-and-lost alternatives of rounds 1-4 were, and their numbers, is in DESIGN.md section 3 and profiles/; their code is gone.
enum SkwSw : int { SW_QUANT_TWIN = 0, SW_DEC_WFRAG, SW_GEMM16W, SW_GEMM16W_NGROUPS, SW_XATTN_FRAG, SW_DECODE_GRAPHS, SW_DECODE_GROUPS, SW_DEC_LN_STATS, SW_PROMPT_PASS,
                   SW_PROMPT_SMALL_GEMM, SW_PROMPT_XATTN_MQ, SW_DEC_AFRAG, SW_DEC_ATTN_FASTV, SW_Q8_LDS, SW_RESAMPLE_SCAN, SW_RESAMPLE_NO_HOST_WALK, SW_COUNT };
int skw_sw(int id);
unsigned skw_sw_epoch();      // bumped by every skw_debug_switch_set: captured step graphs carry the epoch they were built under

enum SkwEpi : int {
    EPI_F32 = 0,        // C f32 [m][n] = (acc + bias[n]) (+ res[m][n])
    EPI_F16_KPERM = 1,  // C f16 [m][kperm(n)] = f16((acc + bias[n]) * scale)
    EPI_GELU_F16_KPERM = 2, // C f16 [m][kperm(n)] = f16(gelu(acc + bias[n]))
    EPI_CONV2 = 3,      // C f32 [m][n] = pe[(m % n_ctx)][n] + gelu(acc + bias[n])
    EPI_HEADS_F16 = 4,  // Q/K for attention: C f16 [(b*H+h)*Tpad + i][kperm(d)], m=(b,i), n=(h,d); f16((acc+bias)*scale)
    EPI_VT_F16 = 5,     // swapped call (m = feature, n = token): C f16 [(b*H+h)*64 + c][kperm(key)] (row stride Tpad); f16(acc + bias[m])
    EPI_F16_PLAIN = 6,  // C f16 [m][n] = f16((acc + bias[n]) * scale)   (decoder K/V caches, cross K/V)
    EPI_DEC_QKV = 8,    // fused decoder q|k|v: see epilogue (C = q plain, C2/C3 = K/V cache rows, n_ctx = d)
    EPI_GELU_F32 = 9,   // C f32 [m][n] = gelu(acc + bias[n]), not rounded (feeds ggml's q8 activation quantiser: skw_kernels_q8.hip)
    EPI_GELU_F16_KPERM_ROWPAD = 7, // conv1: like 2 but row index remapped m -> (m / T) * (T + 2) + (m % T) + 1 (one zero row of padding per clip side)
};

struct SkwGemmArgs {
    const half_t* A; long lda;      // [M][K] f16, K axis kperm'ed, lda in elements (multiple of 8)
    int a_rows_per_batch; long a_batch_stride; // when a_rows_per_batch > 0: row m lives at A + (m / rpb) * a_batch_stride + (m % rpb) * lda
    const half_t* W; long ldw;      // [N][K] f16, K axis kperm'ed
    const half_t* Wf;               // skw_gemm16_small / _lnA only, may be null: the same weight as a FRAGMENT-ORDER image (skw_make_wfrag) — per 16-row strip s and 32-k block kb one
                                    // contiguous KiB [lane r16 + 16 g][8 halves] = W[row(16 s + r16)][32 kb + 8 g ..], so that a wave's weight stream is one contiguous run
                                    // instead of 16 rows x 64 B per load instruction (decode: fc2 9.2 -> 7.3 us per launch with weights from HBM, the others -0.3 .. -0.5)
    int M, N, K;                    // K multiple of 32 (zero padded by the producer)
    void* C; long ldc;
    void* C2; void* C3; long ldc2;   // EPI_DEC_QKV
    const int* pos_ptr; int pos_stride; // EPI_DEC_QKV: row m appends its K/V at cache position pos_ptr[m * pos_stride] (device-side, so a captured step graph is step-invariant)
    const float* bias;              // may be null
    const float* res; long ldres;   // EPI_F32 residual (may be null; may alias C)
    float scale;                    // EPI_F16*/HEADS (1.0f = none; multiply is skipped when has_scale == 0)
    int has_scale;
    const uint16_t* gelu_tab;       // EPI_GELU*, EPI_CONV2
    const float* pe; int n_ctx;     // EPI_CONV2: pe [n_ctx][N]; EPI_HEADS/VT: rows per batch item
    int H; int Tpad;                // EPI_HEADS / EPI_VT
    int epi;
    int c_frag, a_frag;             // the f16 decode kernels (skw_gemm16_small / _lnA): c_frag — an EPI_GELU_F16_KPERM product writes C as the fragment-order A image of the product that follows
                                    // (per 16-row tile and 32-k block one KiB [lane r16 + 16 g][8 halves], skw_afrag_off); a_frag — A is such an image.  The decode step's fc1 -> fc2 pair.
    int frag;                       // skw_gemm16, EPI_F16_PLAIN / EPI_VT_F16 with n_ctx, H, Tpad set: C is the fragment-order cross K / V^T image (skw_kfrag_off / skw_vtfrag_off) instead of rows
    int wgroups;                    // k_gemm16w only, set by its launcher: the XCDs split the features into this many n-tile groups (1: the contiguous walk)
    int probe;                      // measurement only (skw_debug_gemm16): bit 0 skip the K-loop DMA, bit 1 skip the MFMAs, bit 2 skip the epilogue
    const float* ln_x; const float* ln_w; const float* ln_b;   // skw_gemm16_small_lnA: A = LayerNorm(ln_x [M][K] f32; ln_w, ln_b) computed inside the GEMM (A / lda unused)
};

// big-M GEMM (LDS-tiled 128x128 block, 4 waves)
void skw_gemm(const SkwGemmArgs& a, hipStream_t s);
// f16-MFMA form of skw_gemm (skw_kernels_f16.hip): same operands and epilogues, K % 64 == 0
void skw_gemm16(const SkwGemmArgs& a, hipStream_t s);
bool skw_gemm16_takes_w(const SkwGemmArgs& a);      // true: skw_gemm16 launches k_gemm16w (weights from the fragment-order image a.Wf) for these arguments, false: k_gemm16
bool skw_gemm16_small(const SkwGemmArgs& a, hipStream_t s);
// A = LayerNorm(ln_x [M][K] f32; ln_w, ln_b), statistics and normalisation inside the kernel from the rows it holds in registers; W in NATURAL k order
bool skw_gemm16_small_lnA(const SkwGemmArgs& a, hipStream_t s);
void skw_attn_encoder16(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out, int B, int H, int n_ctx, int Tpad, hipStream_t s);
// cross attention of the prompt pass (f16_mfma): the encoder attention kernel with a sequence's prompt tokens as the queries — one read of the sequence's cross K / V^T per 128 of them.
// q: [rows][d] f16 plain (scaled); sequence i: rows row0[i] .. + nq[i], cross K / V^T of window slot slot[i]; out: [rows][kperm(d)] f16
// fragment-order image of a [N][ldw] f16 weight (N % 16 == 0, K % 32 == 0; perm: rows taken in the kperm'ed output order of the GELU epilogues); out: N * K halves
void skw_make_wfrag(const half_t* W, long ldw, int N, int K, int perm, half_t* out, hipStream_t s);
void skw_xattn_prefill16(const half_t* q, const half_t* ck, const half_t* cvt, half_t* out, int n_seq, int nq_max, const int* row0, const int* nq, const int* slot,
                         int H, int d, int n_ctx, int Tpad, hipStream_t s, int frag = 0, int ofrag = 0);
// small-M GEMM (M <= 64): fragments straight from global memory, one 16-column strip per wave
void skw_gemm_smallm(const SkwGemmArgs& a, hipStream_t s);

// LayerNorm over rows of f32 x[rows][d] (ggml_norm + mul + add); out16: f16 kperm'ed (may be null); out32: f32 (may be null)
void skw_layernorm(const float* x, int rows, int d, const float* w, const float* b, half_t* out16, float* out32, hipStream_t s);

// Encoder self-attention, exact three-pass softmax. Qh/Kh: [(b*H+h)*Tpad + i][64 kperm], Vt: [(b*H+h)*64 + c][Tpad kperm],
// out: f16 [b*n_ctx + i][kperm(h*64 + c)] with row stride ld_out
// f32_out: `out` is a float [B*n_ctx][ld_out] buffer, natural order, unrounded
void skw_attn_encoder(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out, int B, int H, int n_ctx, int Tpad, hipStream_t s,
    float* dbg = nullptr, float* dbg2 = nullptr, int f32_out = 0);

// log-mel front end
struct SkwMelTables { const float* hann; const float* sin_t; const float* cos_t; const float* filters; int n_mel; int n_fft_bins;
                      const int* grp_lo; const int* grp_hi; /* per filter: the 4-bin groups [lo, hi) that hold its non-zero taps (may be null: all groups) */ };
// raw log10 mel: mel_raw [b][frame][n_mel] f32 for frames < n_calc[b]; frames beyond get log10(1e-10)
void skw_mel_frames(const float* pcm, const long* pcm_off, const int* n_samples, const int* n_len, int B, int n_len_max, SkwMelTables t, float* mel_raw, hipStream_t s);
// per-clip max -> clamp (max-8) and (x+4)/4, in place; tmp: [B] doubles
void skw_mel_normalize(float* mel, const int* n_len, int B, int n_len_max, int n_mel, float* clip_max, hipStream_t s);
// build conv1's im2col rows for the window starting at seek[b]: out f16 [b*T + t][256 kperm] (k = tap*n_mel + c, zero padded to 256)
void skw_mel_im2col(const float* mel, const int* clip_idx, const int* seek, const int* n_len, int Bw, int n_len_max, int n_mel, int T, int k_pad, half_t* out, hipStream_t s);

// ---------------- decoder ----------------
// x[b][d] = f32(te[tok[b]][kperm(i)]) + pe[pos[b]][i]
void skw_dec_embed(const half_t* te, const float* pe, const int* tok, const int* pos, int B, int d, float* x, hipStream_t s);
// ggml's arithmetic for block-quantised files (skw_kernels_q8.hip)
struct SkwQ8Out { int8_t* q; float* dT; float* sT; int M; };   // where a producer leaves its rows as q8 blocks (q == nullptr: it does not)
struct SkwQ8Args { const int8_t* qa; const float* dyT; const float* syT;        // activations: int8 [M][K], scales [K/32][M]
                   const int8_t* qw; const float* dwT; const float* mwT; int n_pad; int form;
                   // segmented: the decoder's form — four contiguous runs of blocks, partial sums added in ascending order (D3'); needs K % 128 == 0   // weights: int8
                   //  [N][K], scales [K/32][n_pad]; form: skw_ggml_dot_form
                   int segmented; };
void skw_q8_quantize(const float* x, long ldx, int M, int K, int8_t* q, float* dT, float* sT, hipStream_t s);
bool skw_gemm_q8(const SkwGemmArgs& a, const SkwQ8Args& qa, hipStream_t s);
void skw_dec_embed_f32(const float* te32, const float* pe, const int* tok, const int* pos, int B, int d, float* x, hipStream_t s);
void skw_dec_embed_ln(const half_t* te, const float* pe, const int* tok, const int* pos, int B, int d, float* x, const float* w, const float* b, half_t* y16, hipStream_t s);
// self attention for one new token per sequence. q: f16 plain [b][d] (already scaled+rounded), kc/vc: f16 plain [b][n_text_ctx][d];
// n_kv[b] = pos[b]+1. out f16 [b][kperm(d)]
// active: &state[0].active of the rows (stride sizeof(SkwSeqState)), rows whose flag is 0 are skipped; may be null
void skw_dec_self_attn(const half_t* q, const half_t* kc, const half_t* vc, const int* pos, int B, int H, int d, int n_text_ctx, half_t* out, const int* active,
    hipStream_t s, int f32_out = 0, SkwQ8Out q8 = SkwQ8Out{nullptr, nullptr, nullptr, 0},
                       // ofrag: output as a fragment-order A image (skw_afrag_off); fastv: the tolerance precision's P.V (16-byte V pieces, per-lane key shares)   // seq
                       //  (stride of `active`): row b uses the K / V cache of sequence seq[b] (the prompt pass: several rows per sequence); null: sequence b
                       const int* seq = nullptr, int fastv = 0, int ofrag = 0);
// bandwidth form: cross V stored per head transposed, cvt: [(b*H+h)*64 + c][Tpad kperm]
// pv16: P.V on the f16 matrix cores (the f16_mfma precision; the exact one chains f32 MFMAs key by key)
// pv16: 0 the exact precision (two-phase kernel, f32 P.V chains), 1 f16_mfma over the row layouts (two-phase kernel, f16 P.V), 2 f16_mfma over the fragment-order images (one streaming pass)
// In-kernel launch clock of the one-pass kernel (bench.py's roofline line must time the launches of the TIMED configuration — step graphs, two row groups on two streams — where no
// HIP event can sit between captured kernels, and which a profiler serialises).  One SkwKClk per graph node (row group, layer).  Launches of a node are serial on its stream, so
// every workgroup counts its own launches in a cell of its own (cnt[workgroup]: a plain load and store, nobody else touches it) — that number is the launch's record.  At its end,
// thread 0 of every live workgroup folds its entry time into t0 (stored inverted, so zeroed memory is the identity of both maxima) and its exit time into t1, on the device's
// constant-rate counter (wall_clock64, hipDeviceAttributeWallClockRate), in one of SKW_KCLK_SHARDS copies of the record (workgroup % shards: same-address atomics from 256
// workgroups cost microseconds, a sixteenth of them does not); the host reduces the copies.  live_rows counts the rows that streamed.  t1 - t0 is first wave in to last wave out:
// what rocprofv3 calls the duration minus the dispatch and completion-signal edges, which skw_debug_xattn measures (events and clock on the same isolated launches).
#define SKW_KCLK_SHARDS 16
#define SKW_KCLK_MAX_WG 4096
struct SkwKClkRec { unsigned long long t0_inv, t1; unsigned live_rows, pad; };
struct SkwKClk { unsigned cap, pad0, pad1, pad2; unsigned cnt[SKW_KCLK_MAX_WG]; SkwKClkRec rec[1][SKW_KCLK_SHARDS]; };      // rec[cap][SKW_KCLK_SHARDS] follows
void skw_dec_cross_attn_vt(const half_t* q, const half_t* ck, const half_t* cvt, int B, int H, int d, int n_ctx, int Tpad, half_t* out, const int* active, hipStream_t s,
    int f32_out = 0, int pv16 = 0, const int* seq = nullptr,
                           // ofrag (one-pass kernel only): the output rows as the fragment-order A image of the projection that follows; events: stamped at the kernel's
                           //  own begin / end (the engine's per-kernel profile)
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int ofrag = 0, SkwKClk* clk = nullptr);

// per-sequence decoding state kept on the device (whisper_decoder + the bits of whisper_full_with_state's loop that depend on it)
struct SkwSeqState {
    int32_t active;        // still decoding
    int32_t failed, completed;
    int32_t has_ts, seek_delta, result_len;
    int32_t n_tokens;      // sampled tokens so far (i)
    int32_t seek, seek_end;
    int32_t n_prompt;
    float no_speech_prob;
    float min_margin;
    int32_t cur_token;     // token to feed next
    int32_t cur_pos;       // its position
    float temperature;     // 0: argmax; > 0: logits / t, then a std::discrete_distribution draw from the clip's mt19937
    int32_t pad;
};
// The prompt pass (skw_engine.hip, prefill): one SkwSeqState per PROMPT TOKEN, so every kernel of the decode step takes it as a row —
//   active = 1, cur_token / cur_pos = the token and its position, pad = the sequence (window slot) it belongs to, seek = slot * n_text_ctx + position (its K / V cache row).
// whisper_full_with_state: `const int delta_min = 10` mel frames (100 ms) - shortest input transcribed, the loop's stop rule and the decoder's end-of-audio test
#define SKW_DELTA_MIN 10
#define SKW_PROMPT_CAP 240   // [prev] + n_text_ctx/2 past tokens + sot, language, task, notimestamps
#define SKW_RNG_WORDS 625   // std::mt19937 state per clip: mt[624] + index
struct SkwTokenOut { int32_t id, tid; float p, plog, pt, ptsum, margin; };
// one sampling decision as the trace / teacher-forced mode records it (skw_full_batch_traced): what this precision would have chosen, what it was made to
// feed instead (forced_id == chosen_id in a free run), the two largest admissible logits with their owners, the (filtered) logit of the fed token and the
// log-sum-exp of the admissible logits.  Layout == skw_trace_step of include/skw_engine.h.
// temperature > 0: chosen_id is a draw, and the logits are the row's divided by it
struct SkwTraceStep { int32_t chosen_id, forced_id, top1_id, top2_id; float top1, top2, forced_logit, lse; float temperature; int32_t pad; };
struct SkwLogitParams {
    int n_vocab, tok_eot, tok_sot, tok_translate, tok_transcribe, tok_solm, tok_prev, tok_nosp, tok_not, tok_beg;
    int n_lang; int tok_space, tok_sp_dash, tok_sp_quote;
    int suppress_blank, suppress_nst, no_timestamps, single_segment, max_tokens;
    int tid0_initial;   // round(max_initial_ts / precision), <0 disables
    int n_max;          // n_text_ctx/2 - 4
    int any_sampled;    // some row of this pass decodes at a temperature > 0 (host knowledge: the ladder's position per clip): the sampler's draw form (64 KB of LDS staging) is launched only then
};
// whisper_process_logits + whisper_sample_token(best) + the per-token state update of whisper_full_with_state.
// logits: [B][n_vocab] (modified in place), static_mask: [n_vocab] bytes (1 = always suppressed: specials, langs, nst list when enabled)
// static_mask buffer = n_vocab bytes (1 = always suppressed), padded to 16, followed by the same bits transposed for the sampling
// kernel's thread layout (two 64-bit words per thread); build it on the host with skw_static_mask_pack
size_t skw_static_mask_bytes(int n_vocab);
__host__ __device__ inline size_t skw_probs_row_floats(int n_vocab) { return (size_t)3 * ((n_vocab + 1) & ~1); }
void skw_static_mask_pack(const uint8_t* mask, int n_vocab, uint8_t* out);
// probs: [B][skw_probs_row_floats(n_vocab)] workspace (written only by rows with temperature > 0): n_vocab f32 probabilities, then n_vocab f64 normalised ones; rng:
//  [clips][SKW_RNG_WORDS], row b draws from rng[clip_idx[b]];
// n_active: [B] live flags (1 while the row decodes; the kernel stores 0 when it completes or fails) — host-mapped memory in the engine
void skw_dec_sample(float* logits, const uint8_t* static_mask, SkwLogitParams p, SkwSeqState* st, SkwTokenOut* toks /*[B][max_tokens]*/, int max_tok, int B, int* n_active,
                    float* probs, uint32_t* rng, const int* clip_idx, const int* prompt_buf /* [B][SKW_PROMPT_CAP]: row b feeds prompt_buf[b][0 .. n_prompt) before it samples */, hipStream_t s,
                    const int* forced = nullptr /* [B][max_tok]: token to feed after decision i instead of the chosen one (< 0: the chosen one) */, SkwTraceStep* trace = nullptr /* [B][max_tok] */);
void skw_debug_force_stream_sampler(int on);   // tests: the streaming sampler (filters the logits row in place) even where the register-resident form applies
void skw_rng_seed(uint32_t* rng, int n_clips, uint32_t seed, hipStream_t s);   // std::mt19937(seed) for every clip

// ---------------- resampler (R1) ----------------
// start / count / offset: [n_chunks + 1] scratch for the per-chunk proposal; flag: 1 int (set when the proposal had to be redone sequentially)
void skw_resample_linear_launch(const float* in, int channels, double last_index, double t_ratio, int chunk, int n_chunks, int* pos, float* frac, int* n_out, double* last_index_out,
                                // host_proposal: start / count / offset already hold the first proposal
                                float* out, int cap, double* start, int* count, int* offset, int* flag, hipStream_t s, bool host_proposal = false);
void skw_resample_polyphase_launch(const float* in, long in_base, long n_in, long n_total, int channels, const float* coef, int L, int M, int T, float* out, long out_first, long n_out, hipStream_t s);
