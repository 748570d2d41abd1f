// skw_kernels_f16.hip — the "f16_mfma" precision of the encoder's contractions (K2, K4-K6) for gfx950.
//
// Same operands, layouts and epilogues as the exact kernels in skw_kernels.hip; only the contraction differs: the f16 values
// go to the matrix cores as f16 (v_mfma_f32_16x16x32_f16, f32 accumulate, 16x the rate of the f32-input MFMA) instead of
// being widened and chained in k order.  The hardware's internal summation order is not a simple chain (tools/probe/probe_mfma.hip),
// so results are close to, not bit-identical with, the oracle's: the parity bar in this mode is identical token ids / timestamps /
// texts plus stated tolerances on the intermediate tensors (tests/test_gpu_f16.py); the exact mode stays the bit-for-bit checker.
//
// The "kperm" storage order of every contraction axis (inside each aligned block of 32, the 8 values with k % 4 == q are
// contiguous) serves this instruction too: lane group g of a 16x16x32 MFMA takes the 16 bytes at [8g, 8g+8) of a block from
// both operands, i.e. the same eight k values on each side, and a dot product does not care in which slot a k sits.
//
// Reference call site of everything here: /root/reference/plugins/native/whisper/src/lib.rs:644-646 (`whisper_state.full`).
#include "skw_dev_common.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef SKW_VOCAB_W_AUX
#define SKW_VOCAB_W_AUX 0    // (experiment: the vocabulary kernel's weight loads — 80 MB read once per step)
#endif
#ifndef SKW_EPI_ST_NT
#define SKW_EPI_ST_NT 0      // (experiment: the big GEMM's output rows stored with the non-temporal policy, so that they do not push the re-read W / A lines out of L2)
#endif
#ifndef SKW_EPI_RES_NT
#define SKW_EPI_RES_NT 0     // (experiment: the big GEMM's residual rows, read once, with the non-temporal policy)
#endif
#ifndef SKW_DEC_W_AUX
#define SKW_DEC_W_AUX 0      // (experiment: 2 = the decode GEMMs' weight loads carry the non-temporal policy)
#endif
#define MFMA16X32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ------------------------------------------------------------------ big-M GEMM, f16 MFMA
// C[m][n] = sum_k A[m][k] * W[n][k].  Block tile BM x BN, K step 64 (128-byte LDS rows), NWM x NWN waves, each owning a
// (BM/NWM) x (BN/NWN) sub-tile of 16x16 MFMA tiles.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, one
// 1-KiB piece = 8 rows x 128 B per wave-instruction); the LDS image is lane-linear, so the bank swizzle (16-byte chunk c of row
// r sits at chunk position c ^ (r & 7)) is applied to the per-lane SOURCE address and again on the fragment reads.  Two LDS
// buffers, one barrier per K step: the DMA of step k+1 flies under the MFMAs of step k.
//
// Epilogue shape.  The MFMA hands a lane FOUR CONSECUTIVE ROWS of its first operand (rows 4g .. 4g+3) for one column of the
// second.  The operand that supplies those rows ("X") is therefore the one whose index runs along memory in the output: the
// weights (n) for every row-major / per-head output, the tokens for V^T.  When the output's n (or key) axis is stored in kperm
// order, the X tile's LDS row c is loaded from source row perm(c) (a free choice: the DMA source address is per lane), so that
// LDS row order == memory order and each lane owns 4 adjacent output elements: one 8-byte (f16) or 16-byte (f32) store per
// MFMA tile instead of four 2- / 4-byte scatters.  The arithmetic per element is unchanged (bias, scale, GELU table, f16 rounding).
__device__ __forceinline__ int inv_kperm32(int p) { return ((p & 7) << 2) | ((p >> 3) & 3); }      // memory position in a 32-block -> logical index
template <int EPI> struct Epi16 {
    static constexpr bool X_IS_M = (EPI == EPI_VT_F16);                                              // V^T: memory runs along the tokens
    static constexpr bool PERM = (EPI == EPI_F16_KPERM || EPI == EPI_GELU_F16_KPERM || EPI == EPI_GELU_F16_KPERM_ROWPAD || EPI == EPI_HEADS_F16 || EPI == EPI_VT_F16);
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// GELU as ggml's table defines it — f16(0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)))) of the f16-rounded argument — evaluated
// instead of looked up: the 128-KB table costs one 2-byte gather per element (a third of the FC1 GEMM's time at f16 matrix rates).
// tanh through exp2 / rcp: within an f32 ulp or two of libm's, i.e. the same f16 except at rounding boundaries (tolerance mode).
__device__ __forceinline__ float gelu16(float v) {
    const float x = h2f(f2h(v));
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  -2u log2(e) = x (c1 + c2 x^2).  ggml's x <= -10 -> 0 and x >= 10 -> x cases are the
    // formula's own limits (exp2 overflows to inf / underflows to 0), so they need no branches.
    const float z = x * __builtin_fmaf(x * x, -2.88539008177792681472f * 0.79788456080286535588f * 0.044715f, -2.88539008177792681472f * 0.79788456080286535588f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));          // the caller rounds to f16
}
// four adjacent output elements: memory position p0 .. p0+3 along X (p0 % 4 == 0), logical X indices x[0..3], the other index y
template <int EPI>
__device__ __forceinline__ void epi_store4(const SkwGemmArgs& a, int y, int p0, const int (&x)[4], f32x4 v) {
    if (EPI == EPI_F32) {                       // X = n, natural order: x[r] = p0 + r
        if (a.bias) { const f32x4 b = *(const f32x4*)(a.bias + p0); v[0] = v[0] + b[0]; v[1] = v[1] + b[1]; v[2] = v[2] + b[2]; v[3] = v[3] + b[3]; }
        if (a.res) { const f32x4 r = *(const f32x4*)(a.res + (long)y * a.ldres + p0); v[0] = v[0] + r[0]; v[1] = v[1] + r[1]; v[2] = v[2] + r[2]; v[3] = v[3] + r[3]; }
        *(f32x4*)((float*)a.C + (long)y * a.ldc + p0) = v;
    } else if (EPI == EPI_CONV2) {
        const f32x4 b = *(const f32x4*)(a.bias + p0); const f32x4 pe = *(const f32x4*)(a.pe + (long)(y % a.n_ctx) * a.N + p0);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = pe[r] + h2f(f2h(gelu16(v[r] + b[r])));
        *(f32x4*)((float*)a.C + (long)y * a.ldc + p0) = o;
    } else if (EPI == EPI_F16_PLAIN) {
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { float t = v[r]; if (a.bias) t = t + a.bias[p0 + r]; if (a.has_scale) t = t * a.scale; o[r] = f2h(t); }
        *(f16x4*)((half_t*)a.C + (long)y * a.ldc + p0) = o;
    } else if (EPI == EPI_F16_KPERM || EPI == EPI_GELU_F16_KPERM || EPI == EPI_GELU_F16_KPERM_ROWPAD || EPI == EPI_HEADS_F16) {
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = v[r]; if (a.bias) t = t + a.bias[x[r]];
            if (EPI == EPI_F16_KPERM || EPI == EPI_HEADS_F16) { if (a.has_scale) t = t * a.scale; o[r] = f2h(t); }
            else o[r] = f2h(gelu16(t));
        }
        long row;
        if (EPI == EPI_GELU_F16_KPERM_ROWPAD) row = ((long)(y / a.n_ctx) * (a.n_ctx + 2) + (y % a.n_ctx) + 1) * a.ldc + p0;
        else if (EPI == EPI_HEADS_F16) { const int b = y / a.n_ctx, i = y % a.n_ctx; row = ((long)(b * a.H + (p0 >> 6)) * a.Tpad + i) * 64 + (p0 & 63); }
        else if (EPI == EPI_GELU_F16_KPERM && a.c_frag) row = skw_afrag_off(y, p0, a.N);      // (decode step: fc1's output as fc2's fragment-order A image)
        else row = (long)y * a.ldc + p0;
        *(f16x4*)((half_t*)a.C + row) = o;
    } else if (EPI == EPI_VT_F16) {             // X = virtual token row b * Tpad + key (memory position p0 = b * Tpad + key position), y = feature
        const int b = p0 / a.Tpad, kp = p0 % a.Tpad; const float bias = a.bias ? a.bias[y] : 0.0f;
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (x[r] % a.Tpad < a.n_ctx) ? f2h(a.bias ? v[r] + bias : v[r]) : (half_t)0.0f;    // pad keys stay zero
        *(f16x4*)((half_t*)a.C + ((long)(b * a.H + (y >> 6)) * 64 + (y & 63)) * a.Tpad + kp) = o;
    }
}

// ---- staged epilogue.  Per-lane stores of a 16 x 16 MFMA tile reach memory as 16 rows x 32- or 64-byte pieces (measured: 1.2 TB/s
// of f16 output, 2.7 TB/s of f32 read + write — half of every K = 768 GEMM's time).  Instead the finished tile is parked in the LDS
// buffer the K loop no longer needs, as [y][x] rows in output precision (16-byte chunk ck of row y at chunk ck ^ (y & mask): the 16
// rows a wave-instruction writes land on different banks), and all threads then move whole rows: 16 bytes per lane, 512-byte or
// 1-KiB contiguous runs per row, residual / positional-embedding operands read the same way.
template <int EPI> struct Epi16Out { static constexpr bool F32OUT = (EPI == EPI_F32 || EPI == EPI_CONV2); };
// the value an output element takes before the row-wise operands (residual, positional embedding) are added; f16-valued for f16 outputs.
// `bias` is the element's bias (0 when the GEMM has none): the caller takes it from the tile's LDS copy — a global load per element
// here is a dependent round trip each (hipcc waits vmcnt(0) per use while LDS-DMA is in flight) and was 60 % of the epilogue.
template <int EPI>
__device__ __forceinline__ float epi_value(const SkwGemmArgs& a, int xlog, float v, float bias) {
    if (EPI == EPI_F32) { if (a.bias) v = v + bias; return v; }
    if (EPI == EPI_CONV2) { v = v + bias; return h2f(f2h(gelu16(v))); }                      // (f16-valued GELU + f32 positional embedding)
    if (EPI == EPI_VT_F16) { if (a.bias) v = v + bias; return v; }                         // (pad keys are zeroed by the caller, which knows the key index)
    if (a.bias) v = v + bias;
    if (EPI == EPI_GELU_F16_KPERM || EPI == EPI_GELU_F16_KPERM_ROWPAD) return gelu16(v);      // (f16 outputs are rounded once, when they are staged)
    if (a.has_scale) v = v * a.scale;
    return v;
}
// where the 16-byte chunk starting at memory position px of row y lives (element offset into C).  (qb, rb) = (Y0 / n_ctx, Y0 % n_ctx) for
// the row-indexed layouts and (X0 / Tpad, X0 % Tpad) for V^T, computed once per tile: a tile is shorter than a clip, so the quotient
// of any of its rows is qb or qb + 1 — no division per chunk.
template <int EPI>
__device__ __forceinline__ long epi_chunk_offset(const SkwGemmArgs& a, int y, int px, int dy, int dx, int qb, int rb) {
    if (EPI == EPI_GELU_F16_KPERM_ROWPAD || EPI == EPI_HEADS_F16) {
        int i = rb + dy, b = qb; if (i >= a.n_ctx) { i -= a.n_ctx; b += 1; }
        if (EPI == EPI_HEADS_F16) return ((long)(b * a.H + (px >> 6)) * a.Tpad + i) * 64 + (px & 63);
        return ((long)b * (a.n_ctx + 2) + i + 1) * a.ldc + px;
    }
    if (EPI == EPI_VT_F16) { int kp = rb + dx, b = qb; if (kp >= a.Tpad) { kp -= a.Tpad; b += 1; } if (a.frag) return skw_vtfrag_off(b, a.H, a.Tpad, y, kp);
    return ((long)(b * a.H + (y >> 6)) * 64 + (y & 63)) * a.Tpad + kp; }
    if (EPI == EPI_F16_PLAIN) { if (a.frag) { int i = rb + dy, b = qb; if (i >= a.n_ctx) { i -= a.n_ctx; b += 1; } return skw_kfrag_off(b, a.H, a.Tpad, i, px); } }
    return (long)y * a.ldc + px;
}

template <int EPI, int BM, int BN, int NWM, int NWN, bool PROBE = false>      // PROBE: the measurement hooks of tools/gemm16_probe.py (a.probe) compiled in; the product kernel has none
__global__ __launch_bounds__(NWM * NWN * 64, (NWM * NWN) / 4) void k_gemm16(SkwGemmArgs a) {
    constexpr int NW = NWM * NWN, WTM = BM / NWM, WTN = BN / NWN, TM = WTM / 16, TN = WTN / 16;
    constexpr int A_PIECES = BM / 8 / NW, B_PIECES = BN / 8 / NW;      // 1-KiB pieces per wave per K step
    constexpr bool X_IS_M = Epi16<EPI>::X_IS_M, PERM = Epi16<EPI>::PERM;
    constexpr int TX = X_IS_M ? TM : TN, TY = X_IS_M ? TN : TM;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "whole pieces per wave");
    __shared__ __attribute__((aligned(1024))) char lds[2 * (BM + BN) * 128 + 1024];        // two K-step buffers + the tile's bias values
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // V^T: the M axis is walked in virtual rows b * Tpad + key so that 32-key kperm blocks never straddle two clips
    const int Mv = X_IS_M ? (a.M / a.n_ctx) * a.Tpad : a.M;
    const int nbn = (a.N + BN - 1) / BN, nbm = (Mv + BM - 1) / BM, nblk = nbn * nbm;
    const int wr = wave / NWN, wc = wave % NWN, r16 = lane & 15, g = lane >> 4;
    // staging addresses: piece q covers tile rows 8q .. 8q+7; lane -> row 8q + (lane >> 3), chunk position lane & 7
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const half_t* gA[A_PIECES]; const half_t* gB[B_PIECES];
    // persistent workgroups: tile ids blockIdx.x, + gridDim.x, ...  (gridDim.x is a multiple of 8, so all tiles of a workgroup fall
    // into the contiguous run of tiles the XCD-aware remap gives its XCD; the n-tiles of one m-tile run back to back on one XCD, so
    // the A panel is read from HBM once)
    auto tile_origin = [&](int tile, int& m0, int& n0) {
        int q = nblk >> 3, r = nblk & 7, x = tile & 7, y = tile >> 3;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        m0 = (bid / nbn) * BM; n0 = (bid % nbn) * BN;
    };
    auto tile_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            int c = (wave * A_PIECES + i) * 8 + prow; if (X_IS_M && PERM) c = (c & ~31) | inv_kperm32(c & 31);
            int gm = m0 + c; if (PROBE && (a.probe & 256)) gm = c;      // (measurement only: every tile reads the first A panel — what the loop would take with the A operand always in L2)
            // virtual -> real token row (pad rows compute on a copy and store zeros)
            if (X_IS_M) { const int b = gm / a.Tpad, key = gm % a.Tpad; gm = min(b, a.M / a.n_ctx - 1) * a.n_ctx + min(key, a.n_ctx - 1); }
            if (gm > a.M - 1) gm = a.M - 1;             // rows past M are computed on a copy of the last row and never stored
            const long off = a.a_rows_per_batch ? (long)(gm / a.a_rows_per_batch) * a.a_batch_stride + (long)(gm % a.a_rows_per_batch) * a.lda : (long)gm * a.lda;
            gA[i] = a.A + off + pchunk * 8;
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            int c = (wave * B_PIECES + i) * 8 + prow; if (!X_IS_M && PERM) c = (c & ~31) | inv_kperm32(c & 31);
            int gn = n0 + c; if (gn > a.N - 1) gn = a.N - 1;
            gB[i] = a.W + (long)gn * a.ldw + pchunk * 8;
        }
    };
    auto stage = [&](int buf, int kb) {
        char* base = lds + buf * (BM + BN) * 128;
        // (a.probe bits 5 / 6, measurement only: the non-temporal policy on the A / W stream)
        if (PROBE && (a.probe & 32)) {
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gA[i] + kb * 64), (lptr_t)(base + (wave * A_PIECES + i) * 1024), 16, 0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gA[i] + kb * 64), (lptr_t)(base + (wave * A_PIECES + i) * 1024), 16, 0, 0);
        }
        if (PROBE && (a.probe & 64)) {
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gB[i] + kb * 64), (lptr_t)(base + BM * 128 + (wave * B_PIECES + i) * 1024), 16, 0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gB[i] + kb * 64), (lptr_t)(base + BM * 128 + (wave * B_PIECES + i) * 1024), 16, 0, 0);
        }
    };
    // fragment read offsets inside a buffer: row * 128 + ((chunk ^ (row & 7)) << 4), chunk = 4 * khalf + g; row & 7 == r16 & 7
    const int fo0 = ((g ^ (r16 & 7)) << 4), fo1 = fo0 ^ 64;
    const int aoff = (wr * WTM + r16) * 128, boff = BM * 128 + (wc * WTN + r16) * 128;
    const int xoff = X_IS_M ? aoff : boff, yoff = X_IS_M ? boff : aoff;
    const int nk = a.K >> 6;
    const int x_lim = X_IS_M ? Mv : a.N, y_lim = X_IS_M ? a.N : a.M;
    int tile = blockIdx.x;
    if (tile >= nblk) return;
    int m0, n0; tile_origin(tile, m0, n0); tile_sources(m0, n0);
    stage(0, 0);
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 acc[TX][TY];
#pragma unroll
        for (int i = 0; i < TX; ++i)
#pragma unroll
            for (int j = 0; j < TY; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nk; ++kb) {
            const char* base = lds + (kb & 1) * (BM + BN) * 128;
            // The next step is staged unconditionally — the last step re-stages itself into the idle buffer (1/nk more L2 -> LDS traffic) — so that the whole K step is ONE basic block:
            // with the `kb + 1 < nk` branch (and the probe's) in front of the fragment reads, hipcc kept the step's address arithmetic and several accumulators' worth of state live
            // across four code paths and spilled (72-116 B of scratch per lane in three of the eight epilogue variants, none now); same-box A/B per launch: Q/K 175-189 -> 167-177 us,
            // cross K 175-180 -> 167, O-proj 286-290 -> 243-246, FC1 739-746 -> 722, FC2 686 -> 641-648; encode 40.4-41.0 -> 38.8 ms per batch
            //  (profiles/r03g/r03g_gemm16_probe_single_block_k_step.txt)
            if (!PROBE) stage((kb + 1) & 1, min(kb + 1, nk - 1));
            else if (kb + 1 < nk && !(a.probe & 1)) stage((kb + 1) & 1, kb + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {       // (SIMD partners taking the two halves in opposite order — a stagger — measured the same: 28.76 vs 28.63 ms per batch)
                if (PROBE && (a.probe & 2)) break;
                const int fo = kk ? fo1 : fo0;
                f16x8 fx[TX], fy[TY];
#pragma unroll
                for (int t = 0; t < TX; ++t) fx[t] = *(const f16x8*)(base + xoff + t * 2048 + fo);
#pragma unroll
                for (int t = 0; t < TY; ++t) fy[t] = *(const f16x8*)(base + yoff + t * 2048 + fo);
#pragma unroll
                for (int i = 0; i < TX; ++i)
#pragma unroll
                    for (int j = 0; j < TY; ++j) acc[i][j] = MFMA16X32(fx[i], fy[j], acc[i][j]);
            }
            if (kb + 1 < nk) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        }
        // every wave must have read the last K step before the next tile's first step lands in buffer 0 (nk == 1: same buffer; nk odd: the last step's);
        // the last step's re-staging has landed before buffer 1 becomes the epilogue's staging area and buffer 0 the next tile's first step
        if (!PROBE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int X0 = X_IS_M ? m0 : n0, Y0 = X_IS_M ? n0 : m0;                 // tile origin along memory (X) and across it (Y)
        tile += gridDim.x;
        const bool more = tile < nblk;
        if (more) { tile_origin(tile, m0, n0); tile_sources(m0, n0); if (!(PROBE && (a.probe & 1))) stage(0, 0); }      // the next tile's first K step flies under this tile's epilogue (buffer 0)
        if (!(PROBE && (a.probe & 4))) {
            constexpr bool F32OUT = Epi16Out<EPI>::F32OUT;
            constexpr int BX = X_IS_M ? BM : BN, BY = X_IS_M ? BN : BM, WTX = X_IS_M ? WTM : WTN, WTY = X_IS_M ? WTN : WTM;
            constexpr int ESZ = F32OUT ? 4 : 2, ROWB = BX * ESZ, CPR = ROWB / 16;           // bytes per staged row, 16-byte chunks per row
            constexpr int RP = ((BM + BN) * 128) / ROWB < BY ? ((BM + BN) * 128) / ROWB : BY;   // rows per pass: what one LDS buffer holds
            constexpr int NPASS = BY / RP, CE = 16 / ESZ;                                    // elements per chunk
            static_assert(BY % RP == 0 && (CPR & (CPR - 1)) == 0, "whole passes, power-of-two chunks per row");
            char* stg = lds + (BM + BN) * 128;                                               // buffer 1
            float* bias_l = (float*)(lds + 2 * (BM + BN) * 128);                             // bias of the tile's 256 (or 128) features, in memory-position order
            const int wX = X_IS_M ? wr : wc, wY = X_IS_M ? wc : wr;
            const int div = X_IS_M ? a.Tpad : (a.n_ctx > 0 ? a.n_ctx : 1), org = X_IS_M ? X0 : Y0;
            const int qb = org / div, rb = org % div;                                          // (scalar: once per tile)
            {   // features run along X (weights) except for V^T, where they run along Y
                constexpr int NB = X_IS_M ? BY : BX;
                if (tid < NB) { const int pos = (X_IS_M ? Y0 : X0) + tid; const int f = (PERM && !X_IS_M) ? ((pos & ~31) | inv_kperm32(pos & 31)) : pos;
                bias_l[tid] = (a.bias && f < a.N) ? a.bias[f] : 0.0f; }
                __syncthreads();
            }
            for (int pass = 0; pass < NPASS; ++pass) {
                // phase A: this pass's rows, from the waves that hold them.  (RAW = the probe's "no per-element epilogue math": a compile-time copy of the loop —
                // tested per element, the measurement hook was a scalar branch in front of every one of a lane's 128 outputs)
                auto phase_a = [&](auto raw_tag) {
                    constexpr bool RAW = decltype(raw_tag)::value;
#pragma unroll
                for (int j = 0; j < TY; ++j) {
                    const int yl = wY * WTY + j * 16 + r16;
                    if ((wY * WTY + j * 16) / RP != pass) continue;                          // wave-uniform
                    const int ylp = yl - pass * RP;
#pragma unroll
                    for (int i = 0; i < TX; ++i) {
                        const int xl = wX * WTX + i * 16 + 4 * g, p0 = X0 + xl;
                        float o[4];
                        const f32x4 bx = X_IS_M ? (f32x4){bias_l[yl], bias_l[yl], bias_l[yl], bias_l[yl]} : *(const f32x4*)(bias_l + xl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = RAW ? acc[i][j][r] : epi_value<EPI>(a, p0 + r, acc[i][j][r], bx[r]);
                            // V^T: pad keys stay zero (logical key of this memory position)
                            if (X_IS_M) { int key = rb + (((xl + r) & ~31) | inv_kperm32((xl + r) & 31)); if (key >= a.Tpad) key -= a.Tpad; if (key >= a.n_ctx) o[r] = 0.0f; }
                        }
                        const int ck = xl / CE;
                        char* dst = stg + ylp * ROWB + ((ck ^ (ylp & (CPR - 1))) << 4);
                        if (F32OUT) *(f32x4*)dst = (f32x4){o[0], o[1], o[2], o[3]};
                        else *(f16x4*)(dst + ((xl % CE) >= 4 ? 8 : 0)) = (f16x4){f2h(o[0]), f2h(o[1]), f2h(o[2]), f2h(o[3])};
                    }
                }
                };
                if (PROBE && (a.probe & 16)) phase_a(std::true_type{}); else phase_a(std::false_type{});
                __syncthreads();
                // phase B: whole rows out, 16 bytes per lane; the row-wise operands of a batch of chunks are requested together
                constexpr int NCH = RP * CPR / (NW * 64), BATCH = !F32OUT ? 2 : (NCH < 4 ? NCH : 4);     // (f16 outputs have no row-wise operand to wait for)
                static_assert((RP * CPR) % (NW * 64) == 0 && NCH % BATCH == 0, "whole chunks per thread");
#pragma unroll 1
                for (int c0 = 0; c0 < NCH; c0 += BATCH) {
                    long off[BATCH]; f32x4 opnd[BATCH]; int lofs[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        const int cid = tid + (c0 + u) * NW * 64, ylp = cid / CPR, pc = cid % CPR, ck = pc ^ (ylp & (CPR - 1));
                        const int y = Y0 + pass * RP + ylp, px = X0 + ck * CE;
                        lofs[u] = ylp * ROWB + (pc << 4);
                        off[u] = (y >= y_lim || px >= x_lim || (PROBE && (a.probe & 8))) ? -1 : epi_chunk_offset<EPI>(a, y, px, pass * RP + ylp, ck * CE, qb, rb);
                        if (EPI == EPI_F32) opnd[u] = (a.res && off[u] >= 0) ? (SKW_EPI_RES_NT ? __builtin_nontemporal_load((const f32x4*)(a.res + (long)y * a.ldres + px))
                                                                                                : *(const f32x4*)(a.res + (long)y * a.ldres + px))
                                                                              : (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (EPI == EPI_CONV2) { int i = rb + pass * RP + ylp; if (i >= a.n_ctx) i -= a.n_ctx;
                        opnd[u] = (off[u] >= 0) ? *(const f32x4*)(a.pe + (long)i * a.N + px) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    }
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        if (off[u] < 0) continue;
                        if (F32OUT) {
                            f32x4 v = *(const f32x4*)(stg + lofs[u]);
                            if (EPI == EPI_F32) { if (a.res) { v[0] = v[0] + opnd[u][0]; v[1] = v[1] + opnd[u][1]; v[2] = v[2] + opnd[u][2]; v[3] = v[3] + opnd[u][3]; } }
                            else { v[0] = opnd[u][0] + v[0]; v[1] = opnd[u][1] + v[1]; v[2] = opnd[u][2] + v[2]; v[3] = opnd[u][3] + v[3]; }
                            if (SKW_EPI_ST_NT) __builtin_nontemporal_store(v, (f32x4*)((float*)a.C + off[u])); else *(f32x4*)((float*)a.C + off[u]) = v;
                        } else { const u32x4 o16 = *(const u32x4*)(stg + lofs[u]); if (SKW_EPI_ST_NT) __builtin_nontemporal_store(o16, (u32x4*)((half_t*)a.C + off[u]));
                        else *(u32x4*)((half_t*)a.C + off[u]) = o16; }
                    }
                }
                __syncthreads();
            }
        }
        if (!more) break;
    }
}

// per-device state (a host with gpu_device "auto" runs one engine worker thread per GPU in one process): CU count and "dynamic LDS limit raised" flags are keyed by
// the current device; the atomics make the lazy initialisation safe from several threads (the worst case is the same value written twice)
static int skw_cur_device() { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0; return dev; }
static int skw_cu_count() {
    static std::atomic<int> n[64];
    const int dev = skw_cur_device(); int v = n[dev].load(std::memory_order_relaxed);
    if (!v) { hipDeviceProp_t p; v = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; n[dev].store(v, std::memory_order_relaxed); }
    return v;
}
// ------------------------------------------------------------------ big-M GEMM, f16 MFMA, weights from a fragment-order image: k_gemm16w
// Round 4.  k_gemm16's 256 x 256 tile keeps one 8-wave workgroup per CU, all of whose waves are in the same phase: its epilogue (HBM-bound for the f32
// residual outputs, VALU-bound for FC1's GELU) runs with the matrix cores idle — 60 of a Q-shape launch's 174 us.  This form is built so that TWO
// independent workgroups share a CU (4 waves each, 2 waves per SIMD as before): one's epilogue, barriers and staging run under the other's MFMAs.
//   tile 128 (tokens) x 256 (features); a wave owns 128 x 64 (32 MFMA tiles, 128 accumulator registers), the four waves side by side along the features;
//   tokens (A) go through LDS as in k_gemm16 — LDS-DMA pieces of 8 rows x 128 B, XOR swizzle on the source address — in a ring of three 16-KiB K-step
//   slots, one barrier per K step, the DMA of step k + 2 issued after the barrier of step k (it overwrites the slot every wave read in step k - 1);
//   weights never touch LDS: a wave's four 16-feature strips come straight from the weight's FRAGMENT-ORDER image (skw_make_wfrag: per strip and 32-k
//   block one contiguous KiB in MFMA lane order — the decode kernels' format) into registers, one 32-k block ahead of the MFMAs that consume them.
//   A wave's weight columns are its own, so nothing is loaded twice within a workgroup; the weight matrix (1.2 - 4.7 MB) is L2-resident.
// 64 KiB of LDS per workgroup: the ring (48 KiB; its first 32 KiB are the epilogue's staging area once the K loop is over), the tile's bias values.
// Epilogues, layouts and per-element arithmetic are k_gemm16's (the same helpers); the summation order inside an MFMA is the hardware's either way.
// workgroup barrier for LDS hand-overs only: every LDS operation of this wave has completed, but global stores (and loads) stay in flight across it —
// __syncthreads() is a fence too, and hipcc drains vmcnt(0) for it: every epilogue pass then waited for its own stores to reach memory
#define SKW_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int EPI, bool PROBE = false>      // PROBE: measurement hooks (tools/gemm16w_probe.py; a.probe bit 10 no epilogue, bit 12 no MFMAs, bit 13 no weight loads, bit 14 no A staging)
__global__ __launch_bounds__(256, 2) void k_gemm16w(SkwGemmArgs a) {
    constexpr int BM = 128, BN = 256, NW = 4, WTN = 64, TMT = BM / 16, TNT = WTN / 16, SLOT = BM * 128, NSLOT = 3, PARK = 32768;
    constexpr bool X_IS_M = Epi16<EPI>::X_IS_M, PERM = Epi16<EPI>::PERM;
    constexpr int TX = X_IS_M ? TMT : TNT, TY = X_IS_M ? TNT : TMT;
    constexpr int A_PIECES = BM / 8 / NW;                                                   // 1-KiB pieces per wave per K step
    __shared__ __attribute__((aligned(1024))) char lds[65536];                              // ring 3 x 16 KiB (its first 32 KiB double as the waves' epilogue staging) | every feature's bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mv = X_IS_M ? (a.M / a.n_ctx) * a.Tpad : a.M;
    const int nbn = (a.N + BN - 1) / BN, nbm = (Mv + BM - 1) / BM, nblk = nbn * nbm;
    const int r16 = lane & 15, g = lane >> 4;
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const int nk = a.K >> 6, nk32 = a.K >> 5, nstrip = a.N >> 4;
    const half_t* gA[A_PIECES];
    unsigned woff[TNT];                                                                     // byte offset of this lane's 16 bytes in k-block 0 of each of the wave's strips
    // Tile -> (m, n).  Workgroups are dealt to the 8 XCDs round-robin, so tile t runs on XCD t & 7.  Default walk: an XCD takes a contiguous run of tiles, n fastest — the n-tiles
    // of a token panel run together on one XCD.  That keeps A's re-reads local, but an XCD then streams the WHOLE weight through its 4 MiB L2 for every few panels: fine while W
    // fits (768 x 768: 1.2 MB), pathological for FC1's 4.7 MB (PMC, profiles/r04i: 1 556 MB fetched per launch against 152 algorithmic — every W strip comes from the
    // Infinity Cache, at its latency, into a loop whose weight loads have a quarter step of slack).  ng > 1 (a.wgroups): the XCDs split the FEATURES — XCD x owns n-tile group
    // x % ng (its W slice stays in L2) for token-panel range x / ng; a panel is then fetched by ng XCDs instead of one.  Walk length is padded; tiles past the end do nothing.
    const int ng = a.wgroups > 1 ? a.wgroups : 1, ngn = nbn / ng, mparts = 8 / ng, mp = (nbm + mparts - 1) / mparts;
    const int nwalk = ng > 1 ? 8 * mp * ngn : nblk;
    auto tile_origin = [&](int tile, int& m0, int& n0) {
        if (ng > 1) {
            const int x = tile & 7, y = tile >> 3, gsel = x % ng, part = x / ng;
            const int m = part * mp + y / ngn;
            m0 = m < min(nbm, (part + 1) * mp) ? m * BM : -1; n0 = (gsel * ngn + y % ngn) * BN;
            return;
        }
        int q = nblk >> 3, r = nblk & 7, x = tile & 7, y = tile >> 3;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        m0 = (bid / nbn) * BM; n0 = (bid % nbn) * BN;
    };
    auto tile_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            int c = (wave * A_PIECES + i) * 8 + prow; if (X_IS_M && PERM) c = (c & ~31) | inv_kperm32(c & 31);
            int gm = m0 + c;
            if (PROBE && (a.probe & 32768)) gm = c;      // (measurement only: every tile reads the first A panel — the loop with the A operand always in L2)
            if (X_IS_M) { const int b = gm / a.Tpad, key = gm % a.Tpad; gm = min(b, a.M / a.n_ctx - 1) * a.n_ctx + min(key, a.n_ctx - 1); }
            if (gm > a.M - 1) gm = a.M - 1;
            const long off = a.a_rows_per_batch ? (long)(gm / a.a_rows_per_batch) * a.a_batch_stride + (long)(gm % a.a_rows_per_batch) * a.lda : (long)gm * a.lda;
            gA[i] = a.A + off + pchunk * 8;
        }
#pragma unroll
        for (int t = 0; t < TNT; ++t) {
            int s = (n0 + wave * WTN + 16 * t) >> 4; if (s > nstrip - 1) s = nstrip - 1;       // strips past N are computed on a copy of the last one and never stored
            woff[t] = (unsigned)(s * nk32) * 1024u + lane * 16;
        }
    };
    auto stage = [&](int slot, int kb) {
        char* base = lds + slot * SLOT;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gA[i] + kb * 64), (lptr_t)(base + (wave * A_PIECES + i) * 1024), 16, 0, 0);
    };
    // The weight loads are inline asm: beside LDS-DMA in flight hipcc waits vmcnt(0) for any ordinary load result (it drained the ring's DMA at every K step's
    // first MFMA), and it does not count asm loads at all — so BOTH queues are counted by hand below.  A load's result is only touched after the wait that names it.
    auto loadw = [&](f16x8 (&w)[TNT], int kb32) {
#pragma unroll
        for (int t = 0; t < TNT; ++t) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[t]) : "v"(woff[t] + (unsigned)kb32 * 1024u), "s"(a.Wf) : "memory");
    };
    const int fo0 = ((g ^ (r16 & 7)) << 4), fo1 = fo0 ^ 64;
    const int aoff = r16 * 128;
    const int x_lim = X_IS_M ? Mv : a.N, y_lim = X_IS_M ? a.N : a.M;
    // The bias of every feature, in the order the epilogue meets them (memory order of the output: the kperm'ed-output epilogues' position p holds logical feature
    // kperm^-1(p)), once per workgroup into the LDS the ring does not use: a tile's epilogue then takes its lanes' values with ds_reads instead of a round trip to L2.
    constexpr int BIAS_OFF = NSLOT * SLOT, BIAS_MAX = (65536 - BIAS_OFF) / 4;
    const bool bias_lds = a.N <= BIAS_MAX;
    if (bias_lds) {
        float* bl = (float*)(lds + BIAS_OFF);
        for (int p = tid; p < a.N; p += 256) { const int f = (PERM && !X_IS_M) ? ((p & ~31) | inv_kperm32(p & 31)) : p; bl[p] = a.bias ? a.bias[f] : 0.0f; }
        SKW_LDS_BARRIER();
    }
    for (int tile = blockIdx.x; tile < nwalk; tile += gridDim.x) {
        int m0, n0; tile_origin(tile, m0, n0);
        if (m0 < 0) continue;                                                                // (padding of the feature-split walk; uniform per workgroup)
        tile_sources(m0, n0);
        f16x8 w0[TNT], w1[TNT];
        // The K loop is software-pipelined at QUARTER-step granularity with no extra registers: a 32-k half step is two groups of 16 MFMAs — token tiles 0-3 (fragments
        // `lo`) and 4-7 (`hi`) against the wave's four weight strips — and while a group's MFMAs issue, the LDS reads of the NEXT group's fragments are in flight:
        //   step k:  P0  wait W0(k)        MFMA lo(h0)   | read hi(h0)
        //            P1                    MFMA hi(h0)   | read lo(h1), load W0(k + 1)
        //            P2  wait W1(k)        MFMA lo(h1)   | read hi(h1)
        //            P3  barrier B(k + 1), DMA G(k + 3)  MFMA hi(h1)   | read lo(h0 of step k + 1), load W1(k + 1)
        // B(k + 1) sits where every wave has COMPLETED its reads of slot k (the hi(h1) fragments are in registers): after it slot k is free for G(k + 3), and slot k + 1
        // — whose DMA each wave retired for its own pieces at P2 — is visible to all.  Vector-memory queue in issue order (G = 4 DMA pieces, W = 4 weight loads):
        //   G(0) G(1) W0(0) G(2) W1(0) | W0(1) | G(3) W1(1) | W0(2) | G(4) W1(2) | ...
        // P0 needs W0(k): younger are G(k + 2), W1(k) -> vmcnt(8).  P2 needs W1(k): younger is W0(k + 1) -> vmcnt(4), which also retires G(k + 2), three quarters of a
        // step after its issue.  G(k + 1) was retired by step k - 1's P2.
        stage(0, 0);
        stage(1, min(1, nk - 1));
        loadw(w0, 0);
        stage(2, min(2, nk - 1));
        loadw(w1, 1);
        f32x4 acc[TX][TY];
#pragma unroll
        for (int i = 0; i < TX; ++i)
#pragma unroll
            for (int j = 0; j < TY; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                   // G(0) has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        f16x8 lo[4], hi[4];
        auto rd = [&](f16x8 (&f)[4], const char* base, int t0, int fo) {
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = *(const f16x8*)(base + aoff + (t0 + t) * 2048 + fo);
        };
        // 16 MFMAs: token tiles t0 .. t0 + 3 against the four weight strips.  The next group's fragment reads are issued right AFTER the first MFMA: the wait hipcc puts
        // in front of that MFMA (for this group's fragments) then has nothing younger to wait for — issued before it, the look-ahead reads were drained too (lgkmcnt(0)).
        auto mm = [&](const f16x8 (&f)[4], const f16x8 (&w)[TNT], int t0, f16x8 (&nf)[4], const char* nb, int nt0, int nfo) {
            if (PROBE && (a.probe & 4096)) { asm volatile("" :: "v"(f[0]), "v"(f[3]), "v"(w[0])); rd(nf, nb, nt0, nfo); return; }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < TNT; ++q) {
                    if (X_IS_M) acc[t0 + t][q] = MFMA16X32(f[t], w[q], acc[t0 + t][q]);
                    else acc[q][t0 + t] = MFMA16X32(w[q], f[t], acc[q][t0 + t]);
                    if (t == 0 && q == 0) { __builtin_amdgcn_sched_barrier(0); rd(nf, nb, nt0, nfo); __builtin_amdgcn_sched_barrier(0); }
                }
        };
        rd(lo, lds, 0, fo0);
        for (int kb = 0; kb < nk; ++kb) {
            const char* base = lds + (kb % NSLOT) * SLOT;
            const char* nbase = lds + ((kb + 1) % NSLOT) * SLOT;
            // P0
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]) :: "memory");
            mm(lo, w0, 0, hi, base, 4, fo0);
            __builtin_amdgcn_sched_barrier(0);
            // P1
            mm(hi, w0, 4, lo, base, 0, fo1);
            if (!(PROBE && (a.probe & 8192))) loadw(w0, min(2 * kb + 2, nk32 - 1));
            __builtin_amdgcn_sched_barrier(0);
            // P2
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]) :: "memory");
            mm(lo, w1, 0, hi, base, 4, fo1);
            __builtin_amdgcn_sched_barrier(0);
            // P3
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]) :: "memory");
            if (!(PROBE && (a.probe & 16384))) stage(kb % NSLOT, min(kb + 3, nk - 1));
            mm(hi, w1, 4, lo, nbase, 0, fo0);
            if (!(PROBE && (a.probe & 8192))) loadw(w1, min(2 * kb + 3, nk32 - 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" :: "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]));                  // (the last look-ahead read is never consumed)
        // the re-staged copies of the last step have landed and every wave is done with the ring before it becomes the staging area.  The wait NAMES the weight
        // registers: the last step's look-ahead load is never consumed, and a register the compiler considers dead while an asm load is still in flight would be
        // handed to the epilogue and then overwritten by the late data (seen: a wild residual address, a memory fault).
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]) :: "memory");
        __syncthreads();
        const int X0 = X_IS_M ? m0 : n0, Y0 = X_IS_M ? n0 : m0;
        if (PROBE && (a.probe & 1024)) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[TX - 1][TY - 1])); }
        else {
            // Epilogue, per wave and without workgroup barriers: the wave's 128 x 64 sub-tile goes through ITS OWN 16 KiB of the (now idle) ring — written in the
            // MFMA's layout (a lane holds 4 adjacent X positions of one Y), read back as whole rows of the output (128 B of f16 / 256 B of f32 per row, 16 bytes per
            // lane, 1 KiB per wave-instruction) — so the four waves drift apart and one's stores run under another's arithmetic; LDS operations of one wave
            // execute in order, so a wait on lgkmcnt is all the synchronisation a fill needs.  Row-wise operands (residual, positional embedding) are requested
            // before the fill they are added to.  Values and rounding are k_gemm16's (epi_value, f2h): the two kernels' outputs are bit-identical.
            constexpr bool F32OUT = Epi16Out<EPI>::F32OUT;
            constexpr int ESZ = F32OUT ? 4 : 2, CE = 16 / ESZ;
            constexpr int WX = X_IS_M ? BM : WTN, WY = X_IS_M ? WTN : BM;                  // the wave's extent along memory (X) and across it (Y): 64 x 128, V^T 128 x 64
            constexpr int RB = WX * ESZ, CPRW = RB / 16;                                   // bytes and 16-byte chunks per staged row
            constexpr int FILLB = 8192;                                                    // bytes per fill and wave: 8 chunks of 16 bytes per lane (f32 outputs: 8 operand requests in flight)
            constexpr int RH = FILLB / RB < WY ? FILLB / RB : WY, NFILL = WY / RH;         // rows per fill, fills per tile
            constexpr int NCH = RH * CPRW / 64, TYF = RH / 16;                             // chunks per lane per fill, Y tiles per fill
            static_assert(WY % RH == 0 && (RH * CPRW) % 64 == 0 && RH % 16 == 0, "whole fills");
            char* wl = lds + wave * FILLB;
            const int xw = X_IS_M ? 0 : wave * WTN, yw = X_IS_M ? wave * WTN : 0;          // the wave's origin inside the tile
            const int div = X_IS_M ? a.Tpad : (a.n_ctx > 0 ? a.n_ctx : 1), org = X_IS_M ? X0 : Y0;
            const int qb = org / div, rb = org % div;
            // (uniform) the tile lies inside the matrix and inside one clip, and the layout is one of the affine ones: the write-out's fast path
            // (compiled in for the per-head layout only, whose general offset — two levels of division-free but long integer arithmetic per chunk — spilled; the row-major
            //  variants measured slower with two paths: Q/K 151 -> 132 us, FC1 647 -> 702 us per launch, profiles/r04b)
            constexpr bool FASTOFF = (EPI == EPI_HEADS_F16);
            const bool interior = FASTOFF && Y0 + BM <= y_lim && X0 + BN <= x_lim && rb + BM <= div;
            const bool has_pad = X_IS_M && (rb + BM > a.n_ctx);                            // (uniform) V^T: the tile holds pad keys (positions n_ctx .. Tpad of a clip), which stay zero
            // bias of this lane's elements: 4 adjacent X positions per X tile (features, in memory order), or one feature per Y tile for V^T
            f32x4 bx[X_IS_M ? 1 : TX]; float by[X_IS_M ? TY : 1];
            const float* bl = (const float*)(lds + BIAS_OFF);
            if (!X_IS_M) {
#pragma unroll
                for (int i = 0; i < TX; ++i) {
                    const int p0 = X0 + xw + i * 16 + 4 * g;
                    if (bias_lds) bx[i] = (p0 < a.N) ? *(const f32x4*)(bl + p0) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int pos = p0 + r, f = PERM ? ((pos & ~31) | inv_kperm32(pos & 31)) : pos; bx[i][r] = (a.bias && f < a.N) ? a.bias[f] : 0.0f; }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TY; ++j) { const int f = Y0 + yw + j * 16 + r16; by[j] = (f < a.N) ? (bias_lds ? bl[f] : (a.bias ? a.bias[f] : 0.0f)) : 0.0f; }
            }
#pragma unroll
            for (int fill = 0; fill < NFILL; ++fill) {
                // the row-wise operands this lane's chunks of the fill will meet, requested before the fill
                f32x4 opnd[F32OUT ? NCH : 1];
                if (F32OUT) {
#pragma unroll
                    for (int u = 0; u < NCH; ++u) {
                        const int cid = lane + 64 * u, rho = cid / CPRW, pc = cid % CPRW, ck = pc ^ (rho & (CPRW - 1));
                        const int dy = yw + fill * RH + rho, y = Y0 + dy, px = X0 + xw + ck * CE;
                        const bool in = y < y_lim && px < x_lim && !(PROBE && (a.probe & 65536));
                        if (EPI == EPI_F32) opnd[u] = (a.res && in) ? *(const f32x4*)(a.res + (long)y * a.ldres + px) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (EPI == EPI_CONV2) { int i = rb + dy; if (i >= a.n_ctx) i -= a.n_ctx;
                                                opnd[u] = in ? *(const f32x4*)(a.pe + (long)i * a.N + px) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    }
                }
                // fill: the lane's 4 adjacent X positions of row (Y) j * 16 + r16, chunk position XOR-ed with the row so that the 16 rows of a wave-instruction spread over the banks
#pragma unroll
                for (int jj = 0; jj < TYF; ++jj) {
                    const int j = fill * TYF + jj, rho = jj * 16 + r16;
#pragma unroll
                    for (int i = 0; i < TX; ++i) {
                        const int xl = i * 16 + 4 * g;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = epi_value<EPI>(a, 0, acc[i][j][r], X_IS_M ? by[j] : bx[i][r]);
                            if (X_IS_M && has_pad) { int key = rb + (((xl + r) & ~31) | inv_kperm32((xl + r) & 31)); if (key >= a.Tpad) key -= a.Tpad; if (key >= a.n_ctx) o[r] = 0.0f; }
                        }
                        const int ck = xl / CE;
                        char* dst = wl + rho * RB + ((ck ^ (rho & (CPRW - 1))) << 4);
                        if (F32OUT) *(f32x4*)dst = (f32x4){o[0], o[1], o[2], o[3]};
                        // (f16: the lane's 8 bytes are one half of a 16-byte chunk.  The 16 lanes a ds_write_b64 serves together hold rows rho .. rho + 15 at one chunk
                        //  index: the XOR spreads them over 8 chunk positions, two rows each — rows 8 apart then take OPPOSITE halves, and the reader swaps them back:
                        //  without this SQ_LDS_BANK_CONFLICT was 40 % of the f16-output variants' LDS-active cycles, profiles/r04c)
                        else *(f16x4*)(dst + ((((xl % CE) >= 4) != (((rho >> 3) & 1) != 0)) ? 8 : 0)) = (f16x4){f2h(o[0]), f2h(o[1]), f2h(o[2]), f2h(o[3])};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);      // (the write-out's address arithmetic stays below the fill: hoisted above it, it was held in registers the accumulators still need)
                // whole rows out.  A lane's chunks of a fill are RS rows apart: inside the matrix and inside one clip (`interior`, uniform) the row-major and per-head
                // layouts are affine in the row, so chunk u's offset is chunk 0's plus u steps — no bounds tests, no division by the clip length per chunk
                constexpr int RS = 64 / CPRW;
                long off0 = 0; long ostep = 0;
                if (interior) {
                    const int rho = lane / CPRW, pc = lane % CPRW, ck = pc ^ (rho & (CPRW - 1)), dy = yw + fill * RH + rho, dx = xw + ck * CE;
                    off0 = epi_chunk_offset<EPI>(a, Y0 + dy, X0 + dx, dy, dx, qb, rb) - (long)ck * CE;      // (the column part of every layout on this path ends in + position-within-row)
                    ostep = (long)RS * (EPI == EPI_HEADS_F16 ? 64 : a.ldc);
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int cid = lane + 64 * u, rho = cid / CPRW, pc = cid % CPRW, ck = pc ^ (rho & (CPRW - 1));
                    const int dy = yw + fill * RH + rho, dx = xw + ck * CE, y = Y0 + dy, px = X0 + dx;
                    long off;
                    if (interior) off = off0 + u * ostep + ck * CE;
                    else {
                        if (y >= y_lim || px >= x_lim) continue;
                        off = epi_chunk_offset<EPI>(a, y, px, dy, dx, qb, rb);
                    }
                    if (PROBE && (a.probe & 65536)) continue;
                    const char* src = wl + cid * 16;
                    if (F32OUT) {
                        f32x4 v = *(const f32x4*)src;
                        if (EPI == EPI_F32) { if (a.res) { v[0] = v[0] + opnd[u][0]; v[1] = v[1] + opnd[u][1]; v[2] = v[2] + opnd[u][2]; v[3] = v[3] + opnd[u][3]; } }
                        else { v[0] = opnd[u][0] + v[0]; v[1] = opnd[u][1] + v[1]; v[2] = opnd[u][2] + v[2]; v[3] = opnd[u][3] + v[3]; }
                        *(f32x4*)((float*)a.C + off) = v;
                    } else { u32x4 o16 = *(const u32x4*)src; if ((rho >> 3) & 1) o16 = (u32x4){o16[2], o16[3], o16[0], o16[1]}; *(u32x4*)((half_t*)a.C + off) = o16; }
                }
                if (fill + 1 < NFILL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads have the rows before the next fill overwrites them
            }
            // every wave has read its rows back before the next tile's first K steps land in the ring
            SKW_LDS_BARRIER();
        }
    }
}
template <int EPI> static void launch_gemm16w(const SkwGemmArgs& a_in, hipStream_t s) {
    SkwGemmArgs a = a_in;
    const int Mv = Epi16<EPI>::X_IS_M ? (a.M / a.n_ctx) * a.Tpad : a.M;
    const int nbn_ = (a.N + 255) / 256, nbm_ = (Mv + 127) / 128;
    // feature-split walk (k_gemm16w's tile_origin): when the weight does not fit beside the token panels in an XCD's 4 MiB L2.  SKW_GEMM16W_NGROUPS: 0 automatic, 1 off, 2 / 4 / 8 forced
    const int ng_env = skw_sw(SW_GEMM16W_NGROUPS);
    int ng = 1;
    if (ng_env > 1) ng = ng_env;
    else if (ng_env == 0 && !Epi16<EPI>::X_IS_M) { const double wb = 2.0 * a.N * a.K; while (ng < 8 && wb / ng > 2.6e6 && nbn_ % (2 * ng) == 0) ng *= 2; }
    if (ng > 1 && (nbn_ % ng || 8 % ng || Epi16<EPI>::X_IS_M)) ng = 1;
    a.wgroups = ng;
    const int nblk = ng > 1 ? 8 * ((nbm_ + 8 / ng - 1) / (8 / ng)) * (nbn_ / ng) : nbm_ * nbn_;
    const int slots = ((a.probe & 2048) ? 1 : 2) * (skw_cu_count() & ~7);      // (probe bit 11: one workgroup per CU)
    if (a.probe) hipLaunchKernelGGL((k_gemm16w<EPI, true>), dim3(std::min(nblk, slots)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm16w<EPI>), dim3(std::min(nblk, slots)), dim3(256), 0, s, a);
}
// the weights-from-image form applies when the weight has an image, the K axis is whole 64-steps and the feature count whole strips
static bool gemm16w_ok(const SkwGemmArgs& a) {
    return skw_sw(SW_GEMM16W) && a.Wf && !(a.probe & 1023) && (a.K & 63) == 0 && (a.N & 15) == 0 && a.M >= 128;      // (probe bits 0-9 are k_gemm16's; 10+ this kernel's)
}
bool skw_gemm16_takes_w(const SkwGemmArgs& a) { return gemm16w_ok(a); }      // tests: which of the two kernels skw_gemm16 would launch for these arguments
template <int EPI> static void launch_gemm16(const SkwGemmArgs& a_in, hipStream_t s) {
    // tile choice: 256 x 256 (8 waves, one workgroup per CU) when both extents fill it, 128 x 128 (4 waves, two per CU) otherwise;
    // persistent workgroups: one grid slot per resident workgroup (rounded to the 8 XCDs), each walks tile ids slot, slot + grid, ...
    const int Mv = Epi16<EPI>::X_IS_M ? (a_in.M / a_in.n_ctx) * a_in.Tpad : a_in.M;
    const bool big = Mv >= 256 && a_in.N >= 256 && Mv % 256 == 0 && a_in.N % 256 == 0;
    const int cus = skw_cu_count() & ~7;
    const SkwGemmArgs& a = a_in;
    if (big && a.probe) { const int nblk = ((Mv + 255) / 256) * ((a.N + 255) / 256);
    hipLaunchKernelGGL((k_gemm16<EPI, 256, 256, 2, 4, true>), dim3(std::min(nblk, cus)), dim3(512), 0, s, a); }
    else if (big) { const int nblk = ((Mv + 255) / 256) * ((a.N + 255) / 256); hipLaunchKernelGGL((k_gemm16<EPI, 256, 256, 2, 4>), dim3(std::min(nblk, cus)), dim3(512), 0, s, a); }
    else { const int nblk = ((Mv + 127) / 128) * ((a.N + 127) / 128); hipLaunchKernelGGL((k_gemm16<EPI, 128, 128, 2, 2>), dim3(std::min(nblk, 2 * cus)), dim3(256), 0, s, a); }
}
// f16-MFMA form of skw_gemm.  Requirements (every Whisper geometry meets them): K % 64 == 0, N % 32 == 0, ldc / ldres % 4 == 0.
// EPI_VT_F16 is called in the NATURAL orientation here (A = tokens [M][K], W = weights [N][K], bias per n), unlike the exact
// kernel's operand-swapped call: the output is the same V^T image.
void skw_gemm16(const SkwGemmArgs& a, hipStream_t s) {
    if (gemm16w_ok(a)) {
        switch (a.epi) {
            case EPI_F32: launch_gemm16w<EPI_F32>(a, s); return;
            case EPI_F16_KPERM: launch_gemm16w<EPI_F16_KPERM>(a, s); return;
            case EPI_GELU_F16_KPERM: launch_gemm16w<EPI_GELU_F16_KPERM>(a, s); return;
            case EPI_GELU_F16_KPERM_ROWPAD: launch_gemm16w<EPI_GELU_F16_KPERM_ROWPAD>(a, s); return;
            case EPI_CONV2: launch_gemm16w<EPI_CONV2>(a, s); return;
            case EPI_HEADS_F16: launch_gemm16w<EPI_HEADS_F16>(a, s); return;
            case EPI_VT_F16: launch_gemm16w<EPI_VT_F16>(a, s); return;
            case EPI_F16_PLAIN: launch_gemm16w<EPI_F16_PLAIN>(a, s); return;
            default: break;
        }
    }
    switch (a.epi) {
        case EPI_F32: launch_gemm16<EPI_F32>(a, s); break;
        case EPI_F16_KPERM: launch_gemm16<EPI_F16_KPERM>(a, s); break;
        case EPI_GELU_F16_KPERM: launch_gemm16<EPI_GELU_F16_KPERM>(a, s); break;
        case EPI_GELU_F16_KPERM_ROWPAD: launch_gemm16<EPI_GELU_F16_KPERM_ROWPAD>(a, s); break;
        case EPI_CONV2: launch_gemm16<EPI_CONV2>(a, s); break;
        case EPI_HEADS_F16: launch_gemm16<EPI_HEADS_F16>(a, s); break;
        case EPI_VT_F16: launch_gemm16<EPI_VT_F16>(a, s); break;
        case EPI_F16_PLAIN: launch_gemm16<EPI_F16_PLAIN>(a, s); break;
    }
}

// ------------------------------------------------------------------ encoder self-attention, f16 MFMA (K4)
// One workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries (two 16-query MFMA tiles).  K rows and V^T
// rows arrive in 64-key blocks through a double-buffered LDS image shared by the four waves (register staged: one 16-byte chunk
// per thread per operand half, swizzled on the store so the fragment reads are conflict-free), so each K / V byte leaves L2 once
// per 128 queries instead of once per 16 as in the exact kernel.  One pass over the keys, flash-attention style: S^T = K.Q^T per
// block, running row maximum, p = exp2((s - max) * scale * log2 e), row sums, and O^T += V^T . P^T with P^T taken straight from the
// S^T accumulators (MFMA row rho of a 16-key tile holds key 4 * (rho & 3) + (rho >> 2), which is the order the kperm'ed V^T rows
// store their keys in).  Normalisation by the row sum happens once, on O.
#define A16_QB 128
// XP — the prompt pass's cross attention (skw_engine.hip, prefill): the same kernel with the queries of a SEQUENCE'S prompt tokens (rows row0 .. row0 + nq of the pass, plain
// [row][d] f16 in natural k order, as the cross-query GEMM leaves them) against that sequence's cross K (plain rows [key][d], natural order: both operands of the score MFMA
// then agree on which k sits in which slot) and V^T (already this kernel's layout).  One read of a sequence's K / V^T serves up to 128 of its prompt tokens instead of one.
// frag: K / V^T are the fragment-order images (skw_kfrag_off / skw_vtfrag_off)      // per sequence of the pass: first row, rows, window slot
struct SkwXPrefill { const int* row0; const int* nq; const int* slot; long ldq; long k_seq_stride; long ldk; int frag; int ofrag_k; };
template <bool XP, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void k_attn_encoder16(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out,
                                                           int H, int n_ctx, int Tpad, float kq_scale, int qblocks, SkwXPrefill xp) {
    __shared__ __attribute__((aligned(1024))) char lds[2][2][64 * 128];   // [buffer][K | V^T][64 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x; int bid = blockIdx.x;
    { int q = nblk >> 3, r = nblk & 7, x = bid & 7, y = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y; }   // the query blocks of one (batch, head) share an XCD's L2
    long bh = bid / qblocks; const int qb = bid % qblocks;
    const int b = (int)(bh / H), h = (int)(bh % H);
    int n_q = n_ctx; long qrow0 = 0;
    if (XP) { n_q = xp.nq[b]; qrow0 = xp.row0[b]; if (qb * A16_QB >= n_q) return; bh = (long)xp.slot[b] * H + h; }      // (uniform per workgroup)
    const int q0 = qb * A16_QB + wave * 32;
    const int r16 = lane & 15, g = lane >> 4;
    f16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qi = q0 + qt * 16 + r16; if (qi > n_q - 1) qi = n_q - 1;
        const half_t* qp = XP ? Qh + (qrow0 + qi) * xp.ldq + h * 64 + g * 8 : Qh + (bh * Tpad + qi) * 64 + g * 8;
        qf[qt][0] = *(const f16x8*)qp; qf[qt][1] = *(const f16x8*)(qp + 32);
    }
    const long krow = XP ? xp.ldk : 64;                                  // halves between consecutive keys of this head
    const bool frag = XP && xp.frag;
    __amdgpu_buffer_rsrc_t rk = frag ? __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + (long)xp.slot[b] * xp.k_seq_stride + (long)h * (Tpad >> 4) * 1024), 0, (unsigned)((Tpad >> 4) * 2048), 0x00020000)
                              : XP ? __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + (long)xp.slot[b] * xp.k_seq_stride + h * 64), 0, (unsigned)((((long)n_ctx - 1) * krow + 64) * 2), 0x00020000)
                                   : __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + bh * Tpad * 64), 0, (unsigned)(Tpad * 64 * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + bh * 64 * Tpad), 0, (unsigned)(64 * Tpad * 2), 0x00020000);
    // staging: chunk id c = tid + 256 i -> row c >> 3, 16-byte chunk c & 7
    // bank swizzles.  A ds_read_b128 is served in four groups of SIXTEEN lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) —
    // each group one pass over the 64 banks: conflict-free when its sixteen 16-byte slots differ modulo 256 bytes, i.e. in (row & 1, stored chunk position).
    // V^T fragments read row r16, chunk 4 kh + g: chunk ^ (row & 7) does it (k_gemm16's reads, the same shape, count zero conflicts).  K fragments read row
    // kappa(r16) = 4 (r16 & 3) + (r16 >> 2): round 3's swizzle (by the reading lane's index, derived for groups of eight lanes) left lanes 12-15 and 20-23 of a group
    // on the same slots — a two-way conflict on every K read, SQ_LDS_BANK_CONFLICT = 81 % of the kernel's LDS-active cycles (profiles/r03m).  Searched over the XOR
    // swizzles that are linear in the row bits against the real groups (tools/lds_swizzle_search.py): chunk ^ ((row >> 1) & 6) is conflict-free for both k halves.
    // (The 16-byte staging stores go eight contiguous lanes — one row, eight chunks — at a time: conflict-free under any XOR of the chunk index.)
    unsigned st_lds[2], st_ldsk[2], st_k[2], st_v[2]; int st_vkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, row = c >> 3, pos = c & 7;
        st_lds[i] = (unsigned)(row * 128 + ((pos ^ (row & 7)) << 4));
        st_ldsk[i] = (unsigned)(row * 128 + ((pos ^ ((row >> 1) & 6)) << 4));      // (round 3's swizzle had a two-way conflict on every K read: profiles/r04c)
        st_k[i] = (unsigned)((row * krow + pos * 8) * 2);            // + kb * 64 rows; rows past Tpad (XP: past n_ctx) fall outside the descriptor: zeros
        st_v[i] = (unsigned)((row * Tpad + pos * 8) * 2);            // + kb * 64 keys; chunks past Tpad are replaced by zeros below
        // the same 16-byte chunks at their fragment-order addresses: a 64-key block is 8 KiB of either image (K: key tile row >> 4 of the block, row 4 (r & 3) + (r >>
        //  2), d half pos >> 2; V^T: 32-key block pos >> 2, channel tile row >> 4)
        if (frag) {
            const int r = row & 15;
            st_k[i] = (unsigned)((((row >> 4) * 2 + (pos >> 2)) * 1024) + (4 * (r & 3) + (r >> 2) + 16 * (pos & 3)) * 16);
            st_v[i] = (unsigned)((((pos >> 2) * 4 + (row >> 4)) * 1024) + (r + 16 * (pos & 3)) * 16);
        }
        st_vkey[i] = pos * 8;
    }
    const int nkb = (Tpad + 63) >> 6;
    const int kappa = 4 * (r16 & 3) + (r16 >> 2);
    const int k_off = kappa * 128, k_sw = (r16 & 3) << 1, v_off = r16 * 128, v_sw = r16 & 7;      // k_sw = (kappa >> 1) & 6: the stored swizzle of the row this lane reads
    u32x4 sk[2], sv[2];
    auto load_k = [&](int kb) {
#pragma unroll
        // (keys past the end lie outside the descriptor: zeros, masked below)
        for (int i = 0; i < 2; ++i) sk[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, st_k[i] + (unsigned)kb * (frag ? 8192u : (unsigned)(krow * 128)), 0, 0);
    };
    auto load_v = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (kb * 64 + st_vkey[i] < Tpad) ? st_v[i] + (unsigned)kb * (frag ? 8192u : 128u) : 0x7fffff00u, 0, 0);
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(&lds[buf][0][st_ldsk[i]]) = sk[i];
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(&lds[buf][1][st_lds[i]]) = sv[i];
    };
    // S^T for the two query tiles against key tile kt of the current block
    auto scores = [&](const char* kbase, int kt, f32x4& s0, f32x4& s1) {
        const f16x8 k0 = *(const f16x8*)(kbase + kt * 2048 + k_off + ((g ^ k_sw) << 4));
        const f16x8 k1 = *(const f16x8*)(kbase + kt * 2048 + k_off + (((4 | g) ^ k_sw) << 4));
        s0 = MFMA16X32(k0, qf[0][0], ((f32x4){0.f, 0.f, 0.f, 0.f})); s1 = MFMA16X32(k0, qf[1][0], ((f32x4){0.f, 0.f, 0.f, 0.f}));
        s0 = MFMA16X32(k1, qf[0][1], s0); s1 = MFMA16X32(k1, qf[1][1], s1);
    };
    // ---- one pass over the keys with a running row maximum (online softmax): per 64-key block the scores of both query tiles, their
    // block maximum (lane-local over 16 scores, then across the four lanes of a query), and — only when some query's maximum grew, which
    // stops happening after the first few blocks — a rescale of that tile's O^T accumulators and row sum by 2^((m_old - m_new) c).
    const float c1 = kq_scale * 1.44269504088896341f;
    float m0 = -INFINITY, m1 = -INFINITY;
    // the row sums ride on the matrix cores: one more MFMA per 32-key half block with an all-ones first operand gives every lane the sum of the (f16-rounded) probabilities the
    // P.V product multiplies — 32 VALU adds per block (issued as 16 v_pk_add_f32, dear beside MFMAs) become 4 MFMAs, and the normaliser is exactly the sum of what was multiplied:
    // 9.55 -> 9.30 ms of encoder attention per batch in a same-box A/B.  (Starting the score accumulators at minus the running maximum, so that the MFMA hands exp2 its argument
    // without an fma per score, was also tried: a zero accumulator is an inline constant, a non-zero one costs a v_mov per register — the same 32 VALU instructions — and 60 more
    // registers: 9.26 vs 8.65 ms on one box, reverted.)
    f32x4 lacc0 = {0.f, 0.f, 0.f, 0.f}, lacc1 = {0.f, 0.f, 0.f, 0.f};
    f16x8 ones; for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;
    f32x4 oacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) oacc[qt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_k(0); load_v(0); store_k(0); store_v(0); __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const char* kbase = &lds[kb & 1][0][0]; const char* vbase = &lds[kb & 1][1][0];
        if (kb + 1 < nkb) { load_k(kb + 1); load_v(kb + 1); }
        f32x4 s0[4], s1[4];
        float bm0 = -INFINITY, bm1 = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            scores(kbase, kt, s0[kt], s1[kt]);
            if (kb == nkb - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (kb * 64 + kt * 16 + 4 * r + g >= n_ctx) { s0[kt][r] = -INFINITY; s1[kt][r] = -INFINITY; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { bm0 = fmaxf(bm0, s0[kt][r]); bm1 = fmaxf(bm1, s1[kt][r]); }
        }
        bm0 = skw_rows_max_f32(bm0); bm1 = skw_rows_max_f32(bm1);      // (v_permlane swaps: no LDS round trip inside the block loop)
        if (__builtin_amdgcn_ballot_w64(bm0 > m0 || bm1 > m1)) {                // wave-uniform
            const float n0 = fmaxf(m0, bm0), n1 = fmaxf(m1, bm1);
            const float a0 = __builtin_amdgcn_exp2f((m0 - n0) * c1), a1 = __builtin_amdgcn_exp2f((m1 - n1) * c1);   // first block: exp2(-inf) = 0 on empty accumulators
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) { oacc[0][ct][r] *= a0; oacc[1][ct][r] *= a1; }
            for (int r = 0; r < 4; ++r) { lacc0[r] *= a0; lacc1[r] *= a1; }
            m0 = n0; m1 = n1;
        }
        const float mc0 = m0 * c1, mc1 = m1 * c1;
        f16x8 p0[2], p1[2];      // P^T fragments [32-key half of the block]: element 4 * (kt & 1) + r of lane (query, g) = key 16 kt + 4 r + g
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[kt][r], c1, -mc0)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[kt][r], c1, -mc1));   // masked keys: exp2(-inf) = 0
                p0[kt >> 1][(kt & 1) * 4 + r] = (half_t)e0; p1[kt >> 1][(kt & 1) * 4 + r] = (half_t)e1;
            }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) { lacc0 = MFMA16X32(ones, p0[kh], lacc0); lacc1 = MFMA16X32(ones, p1[kh], lacc1); }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f16x8 fv = *(const f16x8*)(vbase + ct * 2048 + v_off + (((kh * 4 + g) ^ v_sw) << 4));
                oacc[0][ct] = MFMA16X32(fv, p0[kh], oacc[0][ct]);
                oacc[1][ct] = MFMA16X32(fv, p1[kh], oacc[1][ct]);
            }
        if (kb + 1 < nkb) { store_k((kb + 1) & 1); store_v((kb + 1) & 1); }
        __syncthreads();
    }
    const float inv0 = 1.0f / lacc0[0], inv1 = 1.0f / lacc1[0];      // (every row of the ones product is the same sum)
    // O^T tiles: lane (query = r16, g) holds channels ct * 16 + 4 g + r
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + r16;
        if (qi < n_q) {
            half_t* op = XP ? out + (qrow0 + qi) * ld_out : out + ((long)b * n_ctx + qi) * ld_out;
            const float inv = qt ? inv1 : inv0;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int pk = skw_kperm(h * 64 + ct * 16 + 4 * g + r);
                if (XP && xp.ofrag_k) out[skw_afrag_off((int)(qrow0 + qi), pk, xp.ofrag_k)] = (half_t)(oacc[qt][ct][r] * inv);
                else op[pk] = (half_t)(oacc[qt][ct][r] * inv); }
        }
    }
}
void skw_attn_encoder16(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out, int B, int H, int n_ctx, int Tpad, hipStream_t s) {
    const int qblocks = (n_ctx + A16_QB - 1) / A16_QB;
    // registers capped at 128 (13 dwords of scratch per lane) so that four workgroups share a CU instead of three: 9.44-9.50 against 9.64-9.77 ms per batch in a same-box A/B
    // (profiles/r04i/r04t); the waves of different workgroups are what overlaps one's exponentials with another's MFMAs (uncapped: 143 registers, three per CU)
    hipLaunchKernelGGL((k_attn_encoder16<false, 4>), dim3(qblocks * H * B), dim3(256), 0, s, Qh, Kh, Vt, out, ld_out, H, n_ctx, Tpad, 1.0f / sqrtf(64.0f), qblocks, SkwXPrefill{});
}
// the prompt pass's cross attention: n_seq sequences, sequence i's queries are rows row0[i] .. row0[i] + nq[i] of q [rows][d] (already scaled, like K), its K / V^T those of window slot slot[i]
void skw_xattn_prefill16(const half_t* q, const half_t* ck, const half_t* cvt, half_t* out, int n_seq, int nq_max, const int* row0, const int* nq, const int* slot,
                         int H, int d, int n_ctx, int Tpad, hipStream_t s, int frag, int ofrag) {
    const int qblocks = (nq_max + A16_QB - 1) / A16_QB;
    const SkwXPrefill xp{row0, nq, slot, (long)d, (long)(frag ? Tpad : n_ctx) * d, (long)d, frag, ofrag ? d : 0};
    hipLaunchKernelGGL(k_attn_encoder16<true>, dim3(qblocks * H * n_seq), dim3(256), 0, s, q, ck, cvt, out, (long)d, H, n_ctx, Tpad, 1.0f, qblocks, xp);
}

// what a wave does with a finished 16 x 16 tile: lane (r16, g) holds rows-of-W 4g .. 4g+3 (four adjacent outputs) of row m
template <int EPI>
__device__ __forceinline__ void gemm16_small_finish(const SkwGemmArgs& a, int m, int p0, f32x4 v, f32x4 pre_res, long pre_po) {
    if (m >= a.M || p0 >= a.N) return;
    if (EPI == EPI_F32) {
        if ((a.ldc & 3) || p0 + 3 >= a.N) {            // logits: ldc = n_vocab is odd and the last strip is ragged
#pragma unroll
            for (int r = 0; r < 4; ++r) if (p0 + r < a.N) { float x = v[r];
            if (a.bias) x = x + a.bias[p0 + r]; if (a.res) x = x + a.res[(long)m * a.ldres + p0 + r]; ((float*)a.C)[(long)m * a.ldc + p0 + r] = x; }
        } else {
            if (a.bias) { const f32x4 b = *(const f32x4*)(a.bias + p0); v[0] = v[0] + b[0]; v[1] = v[1] + b[1]; v[2] = v[2] + b[2]; v[3] = v[3] + b[3]; }
            if (a.res) { if (a.ldres & 3) pre_res = (f32x4){a.res[(long)m * a.ldres + p0], a.res[(long)m * a.ldres + p0 + 1], a.res[(long)m * a.ldres + p0 + 2], a.res[(long)m * a.ldres + p0 + 3]};
                         v[0] = v[0] + pre_res[0]; v[1] = v[1] + pre_res[1]; v[2] = v[2] + pre_res[2]; v[3] = v[3] + pre_res[3]; }
            *(f32x4*)((float*)a.C + (long)m * a.ldc + p0) = v;
        }
    } else if (EPI == EPI_F16_PLAIN) { int x[4] = {p0, p0 + 1, p0 + 2, p0 + 3}; epi_store4<EPI_F16_PLAIN>(a, m, p0, x, v); }
    else if (EPI == EPI_GELU_F16_KPERM) {
        int x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ((p0 + r) & ~31) | inv_kperm32((p0 + r) & 31);
        epi_store4<EPI_GELU_F16_KPERM>(a, m, p0, x, v);
    } else if (EPI == EPI_DEC_QKV) {                   // n in [0, d): q (+bias, *scale); [d, 2d): K cache (*scale); [2d, 3d): V cache (+bias)
        const int d = a.n_ctx; f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { float x = v[r]; if (a.bias) x = x + a.bias[p0 + r]; if (p0 < 2 * d) x = x * a.scale; o[r] = f2h(x); }
        if (p0 < d) *(f16x4*)((half_t*)a.C + (long)m * a.ldc + p0) = o;
        else {
            const long po = pre_po;
            half_t* dst = (p0 < 2 * d) ? (half_t*)a.C2 + (long)m * a.ldc2 + po + (p0 - d) : (half_t*)a.C3 + (long)m * a.ldc2 + po + (p0 - 2 * d);
            *(f16x4*)dst = o;
        }
    }
}

// fragment-order image of a decode weight (SkwGemmArgs::Wf): one 16-byte chunk per thread
__global__ void k_make_wfrag(const half_t* W, long ldw, int N, int K, int perm, half_t* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // chunk id in the image: ((strip * nkb + kb) * 64 + lane)
    const int nkb = K >> 5;
    if (i >= (long)(N >> 4) * nkb * 64) return;
    const int lane = (int)(i & 63), kb = (int)((i >> 6) % nkb), strip = (int)((i >> 6) / nkb), r16 = lane & 15, g = lane >> 4;
    int n = strip * 16 + r16; if (perm) n = (n & ~31) | inv_kperm32(n & 31);
    *(u32x4*)(out + i * 8) = *(const u32x4*)(W + (long)n * ldw + kb * 32 + g * 8);
}
void skw_make_wfrag(const half_t* W, long ldw, int N, int K, int perm, half_t* out, hipStream_t s) {
    const long n = (long)(N >> 4) * (K >> 5) * 64;
    hipLaunchKernelGGL(k_make_wfrag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ldw, N, K, perm, out);
}
// ------------------------------------------------------------------ decode GEMM (M <= 64 rows per block row), f16 MFMA
// Weight-streaming form for the batched single-token step (K8-K10).  One workgroup = one 16-column strip of W x 64 rows of A;
// its four waves split the K axis (each chains its quarter on the matrix cores: K/128 MFMAs per row tile instead of the exact
// kernel's K/4 dependent f32 MFMAs), the partial tiles meet in LDS and wave t finishes row tile t.  As in k_gemm16 the weights
// are the MFMA's first operand, so a lane ends up with four adjacent outputs of one row: 16-byte residual loads and stores.
// Operands come straight from global memory (16-byte buffer loads with hardware range checks; a ring of RD k-blocks in flight).
template <int EPI, int MT, int NW, int RDP = 0>
__global__ __launch_bounds__(64 * NW) void k_gemm16_small(SkwGemmArgs a) {
    constexpr bool PERM = (EPI == EPI_GELU_F16_KPERM);
    constexpr int RD = RDP ? RDP : ((MT == 4) ? 6 : 12);               // k-blocks in flight per wave (RDP = 24: a wave's whole quarter of K = 3072, one round trip instead of two)
    __shared__ f32x4 red[NW][MT][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16, my0 = blockIdx.y * (16 * MT);
    const int r16 = lane & 15, g = lane >> 4;
    const int nkw = (a.K >> 5) / NW, kb_lo = w * nkw;                  // k-blocks (of 32) per wave; host guarantees K % (32 NW) == 0
    int wn = n0 + r16; if (PERM) wn = (wn & ~31) | inv_kperm32(wn & 31);
    __amdgpu_buffer_rsrc_t rw = a.Wf ? __builtin_amdgcn_make_buffer_rsrc((void*)a.Wf, 0, (unsigned)((long)a.N * a.K * 2),
        0x00020000) : __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)((long)a.N * a.ldw * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0,
        a.a_frag ? (unsigned)((long)((a.M + 15) & ~15) * a.K * 2) : (unsigned)(((long)(a.M - 1) * a.lda + a.K) * 2), 0x00020000);
    const unsigned oob = 0x7fffff00u;
    // a.probe (tools/dec_gemm_probe.py only): 1 = no weight loads, 2 = no activation loads (out-of-range offsets: zeros without memory traffic),
    // 4 = no partial-sum exchange and no epilogue, 8 = no stores
    // fragment-order weights (a.Wf): strip blockIdx.x's k-blocks are consecutive KiB, lane l's 16 bytes at l * 16 — the row permutation of the GELU epilogues is in the image
    const bool wfrag = a.Wf != nullptr;
    const unsigned wstep = wfrag ? 1024u : 64u;
    const unsigned wo = (a.probe & 1) ? oob
                      : wfrag ? (n0 + 15 < a.N ? (unsigned)(((long)blockIdx.x * (a.K >> 5) + kb_lo) * 1024 + lane * 16) : oob)
                      : wn < a.N ? (unsigned)(((long)wn * a.ldw + kb_lo * 32 + g * 8) * 2) : oob;
    unsigned ao[MT];
    // a.a_frag: the activations are a fragment-order image (written that way by the product before, SkwGemmArgs::c_frag): a row tile's k-blocks are consecutive KiB
    const bool afrag = a.a_frag != 0; const unsigned astep = afrag ? 1024u : 64u;
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int m = my0 + t * 16 + r16;
    ao[t] = (m < a.M && !(a.probe & 2)) ? (afrag ? (unsigned)(((long)(blockIdx.y * MT + t) * (a.K >> 5) + kb_lo) * 1024 + lane * 16) : (unsigned)(((long)m * a.lda + kb_lo * 32 + g * 8) * 2)) : oob;
    }
    u32x4 fw[RD], fa[RD][MT];
#pragma unroll
    for (int j = 0; j < RD; ++j) {
        const bool in = j < nkw;
        fw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in && wo != oob) ? wo + j * wstep : oob, 0, SKW_DEC_W_AUX);
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[j][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, (in && ao[t] != oob) ? ao[t] + j * astep : oob, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // what the epilogue of THIS wave's row tile needs besides the sums is requested now: a short kernel cannot afford a dependent
    // memory round trip after its last MFMA
    const int em = my0 + w * 16 + r16, ep0 = n0 + 4 * g;      // (waves w >= MT have no row tile to finish)
    const bool e_ok = w < MT && em < a.M && ep0 + 3 < a.N;
    f32x4 pre_res = {0.f, 0.f, 0.f, 0.f}; long pre_po = 0;
    if (EPI == EPI_F32 && a.res && e_ok && !(a.ldres & 3)) pre_res = *(const f32x4*)(a.res + (long)em * a.ldres + ep0);
    if (EPI == EPI_DEC_QKV && a.pos_ptr && w < MT && em < a.M) pre_po = (long)a.pos_ptr[(long)em * a.pos_stride] * a.n_ctx;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb0 = 0; kb0 < nkw; kb0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const f16x8 xw = __builtin_bit_cast(f16x8, fw[j]);
            f16x8 xa[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) xa[t] = __builtin_bit_cast(f16x8, fa[j][t]);
            const int nb = kb0 + j + RD; const bool in = nb < nkw;       // refill the slot just read (zeros past the wave's K range: fma(0, 0, acc) == acc)
            fw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in && wo != oob) ? wo + nb * wstep : oob, 0, SKW_DEC_W_AUX);
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[j][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, (in && ao[t] != oob) ? ao[t] + nb * astep : oob, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = MFMA16X32(xw, xa[t], acc[t]);          // D[n = 4g + r][m = 16 t + r16]
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.probe & 4) { if (acc[0][0] == 12345.678f) ((float*)a.C)[0] = 0.f; return; }
#pragma unroll
    for (int t = 0; t < MT; ++t) red[w][t][lane] = acc[t];
    __syncthreads();
    if (w >= MT) return;
    {
        const int t = w;                                   // wave t finishes row tile t
        f32x4 v = red[0][t][lane];
#pragma unroll
        for (int s = 1; s < NW; ++s) { const f32x4 o = red[s][t][lane]; v[0] = v[0] + o[0]; v[1] = v[1] + o[1]; v[2] = v[2] + o[2]; v[3] = v[3] + o[3]; }   // fixed order: deterministic
        if (!((a.probe & 8) && v[0] != 12345.678f)) gemm16_small_finish<EPI>(a, my0 + t * 16 + r16, n0 + 4 * g, v, pre_res, pre_po);
    }
}
// ------------------------------------------------------------------ decode GEMM whose A operand is LayerNorm(x), normalised in registers
// The f16_mfma precision owes no CPU a summation order, so the LayerNorm launches in front of the decode step's QKV, cross-query and FC1 products
// (5.0 us each for 0.2 MB of work, in a chain of dependent launches) can go.  A workgroup of the plain kernel already loads its rows of A whole — wave w
// takes K quarter w of every row — so here it loads the f32 residual rows instead, keeps them in registers, and gets the rows' statistics from what it
// holds: each lane's sum / sum of squares (f64) over its own values, two shuffles across the four lanes that share a row, one LDS exchange across the
// four waves, one barrier.  Then (x - mean) * rstd * gain + bias — two fmas and one f16 rounding per element — on the way into the MFMA.  No table, no
// atomics, nothing for the producer to do.  (Round 2's in-kernel forms staged an f16 image of the rows through LDS and reduced twice; statistics left by
// the producing GEMM — f64 atomics per row: +2.1 us per producer; per-strip pairs: a 49 KB table per row block for every consumer workgroup — were
// measured this round and lost too: DESIGN.md section 3.)  The weights are a second copy in NATURAL k order, so a lane's eight k are eight consecutive
// floats of x.  One-pass variance in f64 (E[x^2] - mean^2): within an ulp of the two-pass f32 form.
template <int EPI, int MT, int NKW, int NT>      // NT: 16-column strips per workgroup — one normalised A fragment feeds NT MFMAs, so the rows are loaded and converted once per NT strips
__global__ __launch_bounds__(256) void k_gemm16_small_lnA(SkwGemmArgs a) {
    constexpr bool PERM = (EPI == EPI_GELU_F16_KPERM);
    constexpr int NW = 4;
    static_assert(MT * NT <= NW, "one finishing wave per (row tile, strip)");
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ f32x4 red[NW][MT * NT][64];
    __shared__ f64x2 rowst[NW][MT][16];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * (16 * NT), my0 = blockIdx.y * (16 * MT);
    const int r16 = lane & 15, g = lane >> 4;
    const int nkw = (a.K >> 5) / NW, kb_lo = w * nkw;               // host: nkw <= NKW
    const bool wfrag = a.Wf != nullptr;      // fragment-order image of the natural-k weight (see k_gemm16_small)
    __amdgpu_buffer_rsrc_t rw = wfrag ? __builtin_amdgcn_make_buffer_rsrc((void*)a.Wf, 0, (unsigned)((long)a.N * a.K * 2),
        0x00020000) : __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)((long)a.N * a.ldw * 2), 0x00020000);
    const unsigned wstep = wfrag ? 1024u : 64u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.ln_x, 0, (unsigned)((long)a.M * a.K * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.ln_w, 0, (unsigned)(a.K * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.ln_b, 0, (unsigned)(a.K * 4), 0x00020000);
    const unsigned oob = 0x7fffff00u;
    unsigned wo[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { int wn = n0 + 16 * q + r16; if (PERM) wn = (wn & ~31) | inv_kperm32(wn & 31);
    wo[q] = wfrag ? (n0 + 16 * q + 15 < a.N ? (unsigned)(((long)(blockIdx.x * NT + q) * (a.K >> 5) + kb_lo) * 1024 + lane * 16) : oob)
          : wn < a.N ? (unsigned)(((long)wn * a.ldw + kb_lo * 32 + g * 8) * 2) : oob;
    }
    const unsigned ko = (unsigned)((kb_lo * 32 + g * 8) * 4);          // byte offset of this lane's eight k inside an f32 row (x, gain, bias)
    unsigned xo[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int m = my0 + t * 16 + r16; xo[t] = m < a.M ? (unsigned)((long)m * a.K * 4) + ko : oob; }
    u32x4 fx[NKW][MT][2], fw[NKW][NT];
#pragma unroll
    for (int j = 0; j < NKW; ++j) {                                    // the rows first: the statistics wait on them
        const bool in = j < nkw;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < MT; ++t) fx[j][t][h] = __builtin_amdgcn_raw_buffer_load_b128(rx, (in && xo[t] != oob) ? xo[t] + j * 128 + h * 16 : oob, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int q = 0; q < NT; ++q) fw[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rw, (j < nkw && wo[q] != oob) ? wo[q] + j * wstep : oob, 0, SKW_DEC_W_AUX);
    const int ft = w / NT, fq = w % NT;                                // the (row tile, strip) this wave finishes (waves >= MT * NT: none)
    const int em = my0 + ft * 16 + r16;
    f32x4 pre_res = {0.f, 0.f, 0.f, 0.f}; long pre_po = 0;
    if (EPI == EPI_DEC_QKV && a.pos_ptr && w < MT * NT && em < a.M) pre_po = (long)a.pos_ptr[(long)em * a.pos_stride] * a.n_ctx;
    // statistics from the registers (out-of-range loads read as zeros and add nothing)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < NKW; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const f32x4 v = __builtin_bit_cast(f32x4, fx[j][t][h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double x = (double)v[e]; s1 += x; s2 = __builtin_fma(x, x, s2); } }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) rowst[w][t][r16] = (f64x2){s1, s2};
    }
    u32x4 fg[NKW][2], fb[NKW][2];                                       // gain / bias rows: L2 hits, requested under the barrier
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) { fg[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rg, j < nkw ? ko + j * 128 + h * 16 : oob, 0, 0);
        fb[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rb, j < nkw ? ko + j * 128 + h * 16 : oob, 0, 0); }
    __syncthreads();
    float sa[MT], sb[MT];                                               // x -> x * sa + sb = (x - mean) * rstd
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const f64x2 p0 = rowst[0][t][r16], p1 = rowst[1][t][r16], p2 = rowst[2][t][r16], p3 = rowst[3][t][r16];
        const double s1 = (p0[0] + p1[0]) + (p2[0] + p3[0]), s2 = (p0[1] + p1[1]) + (p2[1] + p3[1]);
        const double md = s1 / (double)a.K; double var = s2 / (double)a.K - md * md; if (var < 0.0) var = 0.0;
        const float rstd = 1.0f / sqrtf((float)var + 1e-5f); sa[t] = rstd; sb[t] = -(float)md * rstd;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
        if (j < nkw) {      // (uniform)
            const f32x4 g0 = __builtin_bit_cast(f32x4, fg[j][0]), g1 = __builtin_bit_cast(f32x4, fg[j][1]), b0 = __builtin_bit_cast(f32x4, fb[j][0]), b1 = __builtin_bit_cast(f32x4, fb[j][1]);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const f32x4 x0 = __builtin_bit_cast(f32x4, fx[j][t][0]), x1 = __builtin_bit_cast(f32x4, fx[j][t][1]);
                f16x8 xa;
#pragma unroll
                for (int e = 0; e < 4; ++e) { xa[e] = (half_t)__builtin_fmaf(__builtin_fmaf(x0[e], sa[t], sb[t]), g0[e], b0[e]);
                xa[4 + e] = (half_t)__builtin_fmaf(__builtin_fmaf(x1[e], sa[t], sb[t]), g1[e], b1[e]); }
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[t][q] = MFMA16X32(__builtin_bit_cast(f16x8, fw[j][q]), xa, acc[t][q]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) red[w][t * NT + q][lane] = acc[t][q];
    __syncthreads();
    if (w >= MT * NT) return;
    f32x4 v = red[0][w][lane];
#pragma unroll
    for (int s = 1; s < NW; ++s) { const f32x4 o = red[s][w][lane]; v[0] = v[0] + o[0]; v[1] = v[1] + o[1]; v[2] = v[2] + o[2]; v[3] = v[3] + o[3]; }
    gemm16_small_finish<EPI>(a, my0 + ft * 16 + r16, n0 + 16 * fq + 4 * g, v, pre_res, pre_po);
}
template <int EPI> static bool launch_gemm16_small_lnA(const SkwGemmArgs& a, hipStream_t s) {
    // Workgroup shape: its ingest is what bounds these kernels (one CU takes in ~70 GB/s).  x is f32 — twice the bytes of a ready-made f16 row — and every
    // workgroup of a row block loads and normalises the same rows, so a workgroup takes 16 rows and NT strips: NT MFMAs per normalised fragment.
    // four strips for the QKV and FC1 products (N >= 2048), one for the cross-attention query (measured: profiles/r03b/r03b_lnfold_v*.txt, r04i)
    const int nkw = a.K >> 7, nt = a.N >= 2048 ? 4 : 1;
    const dim3 block(256);
#define SKW_LNA(NKWV, NTV) do { hipLaunchKernelGGL((k_gemm16_small_lnA<EPI, 1, NKWV, NTV>), dim3((a.N + 16 * NTV - 1) / (16 * NTV), (a.M + 15) / 16), block, 0, s, a); return true; } while (0)
    if (nkw <= 6) { if (nt == 4) SKW_LNA(6, 4); SKW_LNA(6, 1); }
    if (nkw <= 12) { if (nt == 4) SKW_LNA(12, 2); SKW_LNA(12, 1); }
#undef SKW_LNA
    return false;
}
bool skw_gemm16_small_lnA(const SkwGemmArgs& a, hipStream_t s) {
    if ((a.K & 127) || !a.ln_x || !a.ln_w || !a.ln_b || a.res) return false;
    switch (a.epi) {
        case EPI_F16_PLAIN: return launch_gemm16_small_lnA<EPI_F16_PLAIN>(a, s);
        case EPI_GELU_F16_KPERM: return launch_gemm16_small_lnA<EPI_GELU_F16_KPERM>(a, s);
        case EPI_DEC_QKV: return launch_gemm16_small_lnA<EPI_DEC_QKV>(a, s);
        default: return false;
    }
}

// ------------------------------------------------------------------ vocabulary projection of the decode step (K10), f16 MFMA
// N = n_vocab is ~52k columns: as 16-column strips of k_gemm16_small that is 3 242 workgroups, each re-reading its rows of A from
// L2 (98 KB for 64 rows: four times the 25 KB of W the strip needs) — 72 us per step, of which 24 us is dispatching the workgroups.
// Here A is stationary: one workgroup per CU copies its 16 MT rows of A into LDS once (row stride K + 8 halves: the sixteen lanes
// of a 16-byte read phase then hit sixteen different bank groups) and walks a contiguous range of strips; a wave takes two adjacent
// strips at a time (one A fragment from LDS feeds both), pairs w, w + 4, ..., with 2 RD k-blocks of W in flight, refilled as they
// are consumed, across pair boundaries.  A lane's chain for one output runs over the whole K in ascending order (no split): every
// row sees the same arithmetic whatever the batch.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));             // n_vocab is odd: a row of logits starts on any 4-byte boundary
template <int MT, int RD, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void k_gemm16_vocab(SkwGemmArgs a, int strips_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_a[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int nk = a.K >> 5, bps = nk / RD;                 // k-blocks of 32 per strip, ring refills per strip pair (host: nk % RD == 0)
    const int rowb = a.K * 2 + 16;
    const int my0 = blockIdx.y * (16 * MT);
    const int n_strips = (a.N + 15) >> 4;
    const int s_lo = blockIdx.x * strips_per_wg, s_hi = min((int)(blockIdx.x + 1) * strips_per_wg, n_strips);
    const int n_pairs = (s_hi - s_lo + 1) >> 1;
    const int np = w < n_pairs ? (n_pairs - w + NWV - 1) / NWV : 0;        // this wave's pairs: w, w + NWV, ...
    const int nblk = np * bps;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)((long)a.N * a.ldw * 2), 0x00020000);
    const unsigned oob = 0x7fffff00u;
    auto w_off = [&](int b, int h) -> unsigned {             // byte offset of this lane's 16 bytes of ring block b (strip h of the pair), k-block 0 of the block
        if (b >= nblk || (a.probe & 1)) return oob;
        const int strip = s_lo + 2 * (w + NWV * (b / bps)) + h, kb0 = (b % bps) * RD, wn = strip * 16 + r16;
        return (strip < s_hi && wn < a.N) ? (unsigned)(((long)wn * a.ldw + kb0 * 32 + g * 8) * 2) : oob;
    };
    u32x4 fw[2][RD];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned o = w_off(0, h);
#pragma unroll
        for (int j = 0; j < RD; ++j) fw[h][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, o != oob ? o + j * 64 : oob, 0, SKW_VOCAB_W_AUX);
    }
    // A -> LDS (16-byte chunks, coalesced, eight in flight per thread; rows past M read as zeros)
    { const int cpr = a.K >> 3, total = 16 * MT * cpr;
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (unsigned)(((long)(a.M - 1) * a.lda + a.K) * 2), 0x00020000);
      for (int q0 = threadIdx.x; q0 < total; q0 += 64 * NWV * 8) {
          u32x4 v[8]; int dsto[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
              const int q = q0 + 64 * NWV * u, row = q / cpr, c = q - row * cpr, m = my0 + row;
              dsto[u] = q < total ? row * rowb + c * 16 : -1;
              v[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, (q < total && m < a.M && !(a.probe & 2)) ? (unsigned)(((long)m * a.lda + c * 8) * 2) : oob, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) if (dsto[u] >= 0) *(u32x4*)(lds_a + dsto[u]) = v[u];
      } }
    __syncthreads();
    f32x4 acc[2][MT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[h][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* la = lds_a + r16 * rowb + g * 16;
    f16x8 xa[2][MT];                                                             // A fragments of k-block j (slot j & 1): read one k-block ahead of their MFMAs
    if (nblk > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) xa[0][t] = __builtin_bit_cast(f16x8, *(const u32x4*)(la + t * 16 * rowb));
    }
    for (int b = 0; b < nblk; ++b) {
        const unsigned on0 = w_off(b + 1, 0), on1 = w_off(b + 1, 1);
        const int kb0 = (b % bps) * RD;
        const int kb_next = ((b + 1) % bps) * RD;                                // (reads past the last block stay inside the image: k-block 0)
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const f16x8 xw0 = __builtin_bit_cast(f16x8, fw[0][j]), xw1 = __builtin_bit_cast(f16x8, fw[1][j]);
            fw[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, on0 != oob ? on0 + j * 64 : oob, 0, SKW_VOCAB_W_AUX);
            fw[1][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, on1 != oob ? on1 + j * 64 : oob, 0, SKW_VOCAB_W_AUX);
            const int kbn = (j + 1 < RD) ? kb0 + j + 1 : kb_next;
#pragma unroll
            for (int t = 0; t < MT; ++t) xa[(j + 1) & 1][t] = __builtin_bit_cast(f16x8, *(const u32x4*)(la + t * 16 * rowb + kbn * 64));
#pragma unroll
            for (int t = 0; t < MT; ++t) { acc[0][t] = MFMA16X32(xw0, xa[j & 1][t], acc[0][t]); acc[1][t] = MFMA16X32(xw1, xa[j & 1][t], acc[1][t]); }   // D[n = 4g + r][m = 16 t + r16]
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((b + 1) % bps == 0) {                                                // the pair is complete: out it goes
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int strip = s_lo + 2 * (w + NWV * (b / bps)) + h, p0 = strip * 16 + 4 * g;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int m = my0 + t * 16 + r16;
                    if (m < a.M && strip < s_hi && (!(a.probe & 8) || acc[h][t][0] == 12345.678f)) {
                        float* dst = (float*)a.C + (long)m * a.ldc + p0;
                        f32x4 x = acc[h][t];
                        if (p0 + 3 < a.N) {
                            if (a.bias) { x[0] = x[0] + a.bias[p0]; x[1] = x[1] + a.bias[p0 + 1]; x[2] = x[2] + a.bias[p0 + 2]; x[3] = x[3] + a.bias[p0 + 3]; }
                            *(f32x4_u*)dst = x;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (p0 + r < a.N) dst[r] = a.bias ? x[r] + a.bias[p0 + r] : x[r];
                        }
                    }
                    acc[h][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
}
template <int MT, int RD, int NWV> static void launch_gemm16_vocab_n(const SkwGemmArgs& a, hipStream_t s) {
    const int lds = 16 * MT * (a.K * 2 + 16);
    static std::atomic<bool> once[64];      // the attribute is per device
    { const int dev = skw_cur_device(); if (!once[dev].load(std::memory_order_acquire)) { hipFuncSetAttribute((const void*)k_gemm16_vocab<MT, RD, NWV>,
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once[dev].store(true, std::memory_order_release); } }
    const int n_strips = (a.N + 15) / 16, rows = (a.M + 16 * MT - 1) / (16 * MT);
    const int slots = std::max(1, skw_cu_count() / rows);
    const int spw = (n_strips + slots - 1) / slots;
    hipLaunchKernelGGL((k_gemm16_vocab<MT, RD, NWV>), dim3((n_strips + spw - 1) / spw, rows), dim3(64 * NWV), lds, s, a, spw);
}
template <int MT, int RD> static void launch_gemm16_vocab(const SkwGemmArgs& a, hipStream_t s) {
    // waves per workgroup: a workgroup walks ~12.7 strips = 6.4 strip pairs at Whisper-small's vocabulary; with four waves some take two pairs and some one
    launch_gemm16_vocab_n<MT, RD, 8>(a, s);      // eight: 21.0 us per launch against 22.5 with four (tools/dec_gemm_probe.py), same arithmetic per output
}
// plain f32 output with optional bias, no residual; false when the geometry is outside what it handles
static bool skw_gemm16_vocab(const SkwGemmArgs& a, hipStream_t s) {
    if (a.epi != EPI_F32 || a.res || (a.K & 127) || a.K > 2048) return false;
    if (a.ln_x) return false;
    const int nk = a.K >> 5;
    const bool mt4 = 64 * (a.K * 2 + 16) <= 150 * 1024;                          // 64 rows of A in LDS (K <= 1024), else 32
#define SKW_VOCAB_RD(RDV) do { if (mt4) launch_gemm16_vocab<4, RDV>(a, s); else launch_gemm16_vocab<2, RDV>(a, s); return true; } while (0)
    if (nk == 24 || nk == 12) SKW_VOCAB_RD(12);                                  // small, tiny
    if (nk == 32 || nk == 16) SKW_VOCAB_RD(16);                                  // medium, base
    if (nk == 40) SKW_VOCAB_RD(20);                                              // large
    SKW_VOCAB_RD(4);
#undef SKW_VOCAB_RD
}

// Rows per workgroup: every workgroup re-reads its row block of A from L2 (the weights are the small operand here), and one CU
// takes in only ~70 GB/s, so fewer rows per workgroup = more workgroups each loading less: 16-row blocks unless told otherwise.
template <int EPI> static void launch_gemm16_small(const SkwGemmArgs& a, hipStream_t s) {
    // N >= 2048 (the QKV and FC1 products; the logits product where the vocabulary kernel does not apply): 64-row blocks, W read once per 64 rows; else 16-row blocks.
    // Measured and lost (profiles/r02c, r03j): 32-row blocks, eight waves per strip (a launch costs ~0.8 us per 1000 waves), a wave's whole K quarter of FC2 in flight at once.
    if (a.N >= 2048) hipLaunchKernelGGL((k_gemm16_small<EPI, 4, 4>), dim3((a.N + 15) / 16, (a.M + 63) / 64), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm16_small<EPI, 1, 4>), dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(256), 0, s, a);
}
// f16-MFMA form of skw_gemm_smallm; returns false when the geometry is outside what it handles (K % 128 != 0): the caller then
// launches the exact kernel, which handles everything.
bool skw_gemm16_small(const SkwGemmArgs& a, hipStream_t s) {
    if (a.K & 127) return false;
    if (a.N >= 8192 && !(a.probe & 16) && skw_gemm16_vocab(a, s)) return true;
    if (a.ln_x) return false;                 // only the vocabulary kernel (and skw_gemm16_small_lnA) normalise their A operand
    switch (a.epi) {
        case EPI_F32: launch_gemm16_small<EPI_F32>(a, s); return true;
        case EPI_F16_PLAIN: launch_gemm16_small<EPI_F16_PLAIN>(a, s); return true;
        case EPI_GELU_F16_KPERM: launch_gemm16_small<EPI_GELU_F16_KPERM>(a, s); return true;
        case EPI_DEC_QKV: launch_gemm16_small<EPI_DEC_QKV>(a, s); return true;
        default: return false;
    }
}
