// skw_kernels.hip — hand-written gfx950 (CDNA4) kernels for the Whisper hot path.
// Compile with -ffp-contract=off: every fused multiply-add in here is explicit (see include/skw_math.h).
//
// What each kernel restates (whisper.cpp routine; reference call site
// /root/reference/plugins/native/whisper/src/lib.rs:644-646 `whisper_state.full`):
//   k_mel_*            log_mel_spectrogram + worker (K1)
//   k_gemm*            ggml_mul_mat with f16 src0 / f16-converted src1, f32 accumulate (K2, K4-K6, K8-K10)
//   k_layernorm        ggml_norm + ggml_mul + ggml_add (K3)
//   k_attn_encoder     KQ = mul_mat(K,Q); soft_max_ext; mul_mat(V, KQ_soft_max) (K4)
//   k_dec_*            whisper_decode_internal pieces (K7-K9), whisper_process_logits + greedy (K11)
#include "skw_kernels.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <cstring>
#include "../../include/skw_math.h"

#include "skw_dev_common.h"

// ------------------------------------------------------------------ big-M GEMM
// C[m][n] = chain_k A[m][k] * W[n][k]; 128x128 block tile, 4 waves (2x2), wave tile 64x64 = 4x4 MFMA 16x16x4 tiles.
// LDS rows are 32 halves (one kperm block) + 8 halves of padding (80 B) -> conflict-free ds_read_b128.
#define G_BM 128
#define G_BN 128
#define G_LDS_ROW 40   // halves
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm(SkwGemmArgs a) {
    __shared__ __attribute__((aligned(16))) half_t lds[2][2][G_BM * G_LDS_ROW];   // [buf][A/B][rows*40]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbn = (a.N + G_BN - 1) / G_BN, nbm = (a.M + G_BM - 1) / G_BM, nblk = nbn * nbm;
    // XCD-aware bijective remap: blocks b and b+8 share an XCD; give each XCD a contiguous run of tiles
    int bid = blockIdx.x;
    { int q = nblk >> 3, r = nblk & 7, x = bid & 7, y = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y; }
    const int bm = bid / nbn, bn = bid % nbn;       // n-tiles of one m-tile adjacent -> A panel reused from L2
    const int m0 = bm * G_BM, n0 = bn * G_BN;
    const int wr = wave >> 1, wc = wave & 1, r16 = lane & 15, kq = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: 512 16-byte chunks per operand tile, 2 per thread per operand; buffer loads with hardware range checking (rows past
    // M / N read as zeros, no branches), per-thread byte offsets fixed up front, the k-block offset in an SGPR
    u32x4 stA[2], stB[2];
    const int nk = a.K >> 5;
    const long a_rows = a.a_rows_per_batch ? (long)((a.M - 1) / a.a_rows_per_batch) * a.a_batch_stride + (long)((a.M - 1) % a.a_rows_per_batch) * a.lda : (long)(a.M - 1) * a.lda;
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (unsigned)((a_rows + a.K) * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)(((long)(a.N - 1) * a.ldw + a.K) * 2), 0x00020000);
    unsigned voA[2], voB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + 256 * i, row = c >> 2, kc = c & 3;
        int gm = m0 + row, gn = n0 + row;
        long aoff = a.a_rows_per_batch ? (long)(gm / a.a_rows_per_batch) * a.a_batch_stride + (long)(gm % a.a_rows_per_batch) * a.lda : (long)gm * a.lda;
        voA[i] = (gm < a.M) ? (unsigned)((aoff + kc * 8) * 2) : 0x7fffff00u;
        voB[i] = (gn < a.N) ? (unsigned)(((long)gn * a.ldw + kc * 8) * 2) : 0x7fffff00u;
    }
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            stA[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA[i], kb * 64, 0);
            stB[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, voB[i], kb * 64, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            *(u32x4*)(&lds[buf][0][row * G_LDS_ROW + kc * 8]) = stA[i];
            *(u32x4*)(&lds[buf][1][row * G_LDS_ROW + kc * 8]) = stB[i];
        }
    };
    gload(0); lstore(0); __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nk) gload(kb + 1);
        H8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t].u = *(const uint4*)(&lds[buf][0][(wr * 64 + t * 16 + r16) * G_LDS_ROW + kq * 8]);
            fb[t].u = *(const uint4*)(&lds[buf][1][(wc * 64 + t * 16 + r16) * G_LDS_ROW + kq * 8]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { av[t] = h2f(fa[t].h[e]); bv[t] = h2f(fb[t].h[e]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(av[i], bv[j], acc[i][j]);
        }
        if (kb + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m0 + wr * 64 + i * 16 + kq * 4 + r, n = n0 + wc * 64 + j * 16 + r16;
                if (m < a.M && n < a.N) epi_store<EPI>(a, m, n, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------ small-M GEMM (decode, M <= 64), exact precision
// Weight-streaming form: fragments straight from global memory, a ring of SM_DEPTH k-blocks (16 B per lane per operand per block) in flight to hide HBM latency.
// The contraction runs in FOUR contiguous K segments (D3'): a workgroup is one 16-column strip x one 16-row tile, wave s
// chains segment s (k-blocks [s nk/4, (s+1) nk/4), k-ascending from zero), the partial tiles meet in LDS and wave 0 adds them in ascending
// segment order, ((s0 + s1) + s2) + s3 — oracle/skw_oracle.c gemm_chain_seg4, bit for bit.  A quarter of the dependent-MFMA chain per wave.
template <int EPI, int SM_DEPTH>
__global__ __launch_bounds__(256) void k_gemm_smallm_seg(SkwGemmArgs a) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16, mt = blockIdx.y;
    const int r16 = lane & 15, kq = lane >> 4;
    const int gn = n0 + r16, gm = mt * 16 + r16;
    const unsigned wbytes = (unsigned)((long)a.N * a.ldw * 2), abytes = (unsigned)(((long)(a.M - 1) * a.lda + a.K) * 2);
    // fragment-order weight image (SkwGemmArgs::Wf, built for the f16 decode kernels): the same sixteen bytes per lane from one contiguous KiB per (strip, k-block) instead of 16 rows x 64 B —
    // the values, and so every chain, are unchanged.  (Not for the GELU product: its image holds the rows in the f16 kernels' output order.)
    const bool wfrag = a.Wf != nullptr && EPI != EPI_GELU_F16_KPERM && !(a.N & 15);
    const unsigned wstep = wfrag ? 1024u : 64u;
    __amdgpu_buffer_rsrc_t rw = wfrag ? __builtin_amdgcn_make_buffer_rsrc((void*)a.Wf, 0, (unsigned)((long)a.N * a.K * 2),
        0x00020000) : __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, abytes, 0x00020000);
    const unsigned oob = 0x7fffff00u;
    const int nkq = (a.K >> 5) >> 2, kb_lo = wave * nkq;                 // host guarantees K % 128 == 0
    const unsigned wo = wfrag ? (unsigned)(((long)blockIdx.x * (a.K >> 5) + kb_lo) * 1024 + lane * 16) : (gn < a.N) ? (unsigned)(((long)gn * a.ldw + kb_lo * 32 + kq * 8) * 2) : oob;
    const unsigned ao = (gm < a.M) ? (unsigned)(((long)gm * a.lda + kb_lo * 32 + kq * 8) * 2) : oob;
    H8v fw[SM_DEPTH], fa[SM_DEPTH];
#pragma unroll
    for (int j = 0; j < SM_DEPTH; ++j) {
        fw[j].v = __builtin_amdgcn_raw_buffer_load_b128(rw, (wo == oob || j >= nkq) ? oob : wo + j * wstep, 0, 0);
        fa[j].v = __builtin_amdgcn_raw_buffer_load_b128(ra, (ao == oob || j >= nkq) ? oob : ao + j * 64, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int en = n0 + r16; const bool en_ok = en < a.N && wave == 0;
    float pre_bias = 0.0f, pre_res[4] = {0.f, 0.f, 0.f, 0.f}; long pre_po[4] = {0, 0, 0, 0};
    if (EPI != EPI_VT_F16 && a.bias && en_ok) pre_bias = a.bias[en];
    if (EPI == EPI_F32 && a.res && en_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int m = mt * 16 + kq * 4 + r; if (m < a.M) pre_res[r] = a.res[(long)m * a.ldres + en]; }
    }
    if (EPI == EPI_DEC_QKV && a.pos_ptr && wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int m = mt * 16 + kq * 4 + r; if (m < a.M) pre_po[r] = (long)a.pos_ptr[(long)m * a.pos_stride] * a.n_ctx; }
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb0 = 0; kb0 < nkq; kb0 += SM_DEPTH) {
#pragma unroll
        for (int j = 0; j < SM_DEPTH; ++j) {
            float xa[8], xw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xa[e] = h2f(fa[j].h[e]); xw[e] = h2f(fw[j].h[e]); }
            __builtin_amdgcn_sched_barrier(0);
            const int nb = kb0 + j + SM_DEPTH;
            fw[j].v = __builtin_amdgcn_raw_buffer_load_b128(rw, (wo == oob || nb >= nkq) ? oob : wo + nb * wstep, 0, 0);
            fa[j].v = __builtin_amdgcn_raw_buffer_load_b128(ra, (ao == oob || nb >= nkq) ? oob : ao + nb * 64, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb0 + j < nkq) {                                         // (uniform) blocks past the segment are skipped, not multiplied by zeros
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = MFMA16(xa[e], xw[e], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    {
        const f32x4 s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) { float v = acc[r] + s1[r]; v = v + s2[r]; v = v + s3[r]; acc[r] = v; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int m = mt * 16 + kq * 4 + r, n = n0 + r16;
        if (!(m < a.M && n < a.N)) continue;
        float v = acc[r];
        if (EPI == EPI_F32) {
            if (a.bias) v = v + pre_bias;
            if (a.res) v = v + pre_res[r];
            ((float*)a.C)[(long)m * a.ldc + n] = v;
        } else if (EPI == EPI_F16_PLAIN) {
            if (a.bias) v = v + pre_bias;
            if (a.has_scale) v = v * a.scale;
            ((half_t*)a.C)[(long)m * a.ldc + n] = f2h(v);
        } else if (EPI == EPI_GELU_F16_KPERM) {
            if (a.bias) v = v + pre_bias;
            ((half_t*)a.C)[(long)m * a.ldc + skw_kperm(n)] = f2h(gelu_dev(v, a.gelu_tab));
        } else if (EPI == EPI_DEC_QKV) {
            const int d = a.n_ctx;
            if (a.bias) v = v + pre_bias;
            if (n < 2 * d) v = v * a.scale;
            if (n < d) ((half_t*)a.C)[(long)m * a.ldc + n] = f2h(v);
            else if (n < 2 * d) ((half_t*)a.C2)[(long)m * a.ldc2 + pre_po[r] + (n - d)] = f2h(v);
            else ((half_t*)a.C3)[(long)m * a.ldc2 + pre_po[r] + (n - 2 * d)] = f2h(v);
        } else epi_store<EPI>(a, m, n, acc[r]);
    }
}

template <int EPI> static void launch_gemm(const SkwGemmArgs& a, hipStream_t s) {
    int nbn = (a.N + G_BN - 1) / G_BN, nbm = (a.M + G_BM - 1) / G_BM;
    hipLaunchKernelGGL(k_gemm<EPI>, dim3(nbn * nbm), dim3(256), 0, s, a);
}
void skw_gemm(const SkwGemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_F32: launch_gemm<EPI_F32>(a, s); break;
        case EPI_F16_KPERM: launch_gemm<EPI_F16_KPERM>(a, s); break;
        case EPI_GELU_F16_KPERM: launch_gemm<EPI_GELU_F16_KPERM>(a, s); break;
        case EPI_GELU_F16_KPERM_ROWPAD: launch_gemm<EPI_GELU_F16_KPERM_ROWPAD>(a, s); break;
        case EPI_CONV2: launch_gemm<EPI_CONV2>(a, s); break;
        case EPI_HEADS_F16: launch_gemm<EPI_HEADS_F16>(a, s); break;
        case EPI_VT_F16: launch_gemm<EPI_VT_F16>(a, s); break;
        case EPI_F16_PLAIN: launch_gemm<EPI_F16_PLAIN>(a, s); break;
    }
}
// K % 128 == 0 (skw_model_load rejects any other decoder width: the oracle's segmented chain, gemm_chain_seg4, is defined for it)
template <int EPI> static void launch_gemm_small(const SkwGemmArgs& a, hipStream_t s) {
    const int nk = a.K >> 5, nkq = nk >> 2;
    const dim3 gs((a.N + 15) / 16, (a.M + 15) / 16);
    if (nkq % 24 == 0) hipLaunchKernelGGL((k_gemm_smallm_seg<EPI, 24>), gs, dim3(256), 0, s, a);
    else if (nkq <= 6) hipLaunchKernelGGL((k_gemm_smallm_seg<EPI, 6>), gs, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm_smallm_seg<EPI, 8>), gs, dim3(256), 0, s, a);
}
void skw_gemm_smallm(const SkwGemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_F32: launch_gemm_small<EPI_F32>(a, s); break;
        case EPI_F16_KPERM: launch_gemm_small<EPI_F16_KPERM>(a, s); break;
        case EPI_GELU_F16_KPERM: launch_gemm_small<EPI_GELU_F16_KPERM>(a, s); break;
        case EPI_F16_PLAIN: launch_gemm_small<EPI_F16_PLAIN>(a, s); break;
        case EPI_DEC_QKV: launch_gemm_small<EPI_DEC_QKV>(a, s); break;
        default: break;
    }
}

// attention outputs: f16 in kperm order for the f16-weight path; f32 in natural order (the same buffer pointer, viewed as float) when the
// projection that follows multiplies by block-quantised weights and ggml quantises the UNROUNDED f32 row (skw_kernels_q8.hip)
__device__ __forceinline__ void att_store(half_t* out, long row_off, int col, float v, int f32_out) {
    if (f32_out) ((float*)out)[row_off + col] = v; else out[row_off + skw_kperm(col)] = f2h(v);
}
// the decode step's attention outputs when the projection that follows reads a fragment-order A image (SkwGemmArgs::a_frag, skw_afrag_off): fk = its K (0: rows)
__device__ __forceinline__ void att_store_m(half_t* out, int m, long ldo, int col, float v, int f32_out, int fk) {
    if (fk && !f32_out) out[skw_afrag_off(m, skw_kperm(col), fk)] = f2h(v); else att_store(out, (long)m * ldo, col, v, f32_out);
}
// ------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ double wave_sum_f64(double v) { return skw_wave_sum_f64(v); }   // DPP + readlane, no LDS crossbar (skw_dev_common.h)
// one wave per R rows; d <= 64 NC (NC = 12 for every width up to Whisper-small's, 24 up to 1536); FULL: d == 64 NC (Whisper-small: straight-line code, no tail predicates).
// R = 2 for the encoder's 96 000-row launches: the kernel is a stream (442 MB per launch) and a wave with one 3 KiB row in flight leaves the CU short of bytes in flight;
// the rows' arithmetic is skw_ln_rows' either way (same bits; profiles/r04i: 0.085 vs 0.11 ms per launch).
template <int NC, bool FULL, int R = 1>
__global__ __launch_bounds__(256) void k_layernorm(const float* x, int rows, int d, const float* w, const float* b, half_t* out16, float* out32) {
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    float v[R][NC], wv[NC], bv[NC];
    bool live[R]; half_t* o16[R]; float* o32[R];
    // the gain / bias loads go out together with the rows (a 64-row decode launch is three dependent round trips otherwise)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r; live[r] = row < rows;
        const float* xr = x + (long)(live[r] ? row : row0) * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const int i = lane + 64 * c; v[r][c] = (FULL || i < d) ? xr[i] : 0.0f; }
        o16[r] = out16 ? out16 + (long)(live[r] ? row : row0) * d : nullptr; o32[r] = out32 ? out32 + (long)(live[r] ? row : row0) * d : nullptr;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) { const int i = lane + 64 * c; const bool in = FULL || i < d; wv[c] = in ? w[i] : 0.0f; bv[c] = in ? b[i] : 0.0f; }
    skw_ln_rows<R, NC, FULL>(v, wv, bv, d, lane, live, o16, o32);
}
void skw_layernorm(const float* x, int rows, int d, const float* w, const float* b, half_t* out16, float* out32, hipStream_t s) {
    const bool two = rows >= 8192 && d == 768;
    const dim3 g(two ? (rows + 7) / 8 : (rows + 3) / 4);
    if (two) hipLaunchKernelGGL((k_layernorm<12, true, 2>), g, dim3(256), 0, s, x, rows, d, w, b, out16, out32);
    else if (d == 768) hipLaunchKernelGGL((k_layernorm<12, true>), g, dim3(256), 0, s, x, rows, d, w, b, out16, out32);
    else if (d <= 768) hipLaunchKernelGGL((k_layernorm<12, false>), g, dim3(256), 0, s, x, rows, d, w, b, out16, out32);
    else hipLaunchKernelGGL((k_layernorm<24, false>), g, dim3(256), 0, s, x, rows, d, w, b, out16, out32);
}

// ------------------------------------------------------------------ encoder attention (three-pass exact softmax)
#define AT_KROW 72   // halves per K row in LDS (64 + 8 pad = 144 B)
#define AT_VROW 40   // halves per V^T row in LDS (32 + 8 pad = 80 B)
__global__ __launch_bounds__(256, 2) void k_attn_encoder(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out,
                                                         int H, int n_ctx, int Tpad, float kq_scale, float* dbg, float* dbg2, int f32_out) {
    __shared__ __attribute__((aligned(16))) half_t ldsK[2][32 * AT_KROW];
    __shared__ __attribute__((aligned(16))) half_t ldsV[2][64 * AT_VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const long bh = (long)b * H + h;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int r16 = lane & 15, g = lane >> 4;
    const bool wave_on = q0 < n_ctx;
    // Q fragments (B operand): lane (q = r16, kq = g) holds Q[q][d = 32*blk + 4*e + g]
    float qf[2][16];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qi = q0 + qt * 16 + r16; if (qi >= n_ctx) qi = n_ctx - 1;
        const half_t* qp = Qh + (bh * Tpad + qi) * 64 + g * 8;
        H8 x0, x1; x0.u = *(const uint4*)qp; x1.u = *(const uint4*)(qp + 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) { qf[qt][e] = h2f(x0.h[e]); qf[qt][8 + e] = h2f(x1.h[e]); }
    }
    const int nkb = Tpad >> 5;
    // staging maps
    const int krow = tid >> 3, kcc = tid & 7;   // K: 32 rows x 8 chunks
    const int vrow = tid >> 2, vcc = tid & 3;   // V^T: 64 rows x 4 chunks
    const half_t* kg = Kh + (bh * Tpad + krow) * 64 + kcc * 8;
    const half_t* vg = Vt + (bh * 64 + vrow) * Tpad + vcc * 8;
    // MFMA row rho (= r16) of a 16-key tile holds key kappa = 4*(rho&3) + (rho>>2)
    const int kappa = 4 * (r16 & 3) + (r16 >> 2);

    float rmax[2] = {-INFINITY, -INFINITY};
    double rsum[2] = {0.0, 0.0};
    float rinv[2] = {0.f, 0.f};
    f32x4 oacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) oacc[qt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int pass = 0; pass < 3; ++pass) {
        uint4 stK, stV;
        stK = *(const uint4*)kg; if (pass == 2) stV = *(const uint4*)vg;
        *(uint4*)(&ldsK[0][krow * AT_KROW + kcc * 8]) = stK;
        if (pass == 2) *(uint4*)(&ldsV[0][vrow * AT_VROW + vcc * 8]) = stV;
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) { stK = *(const uint4*)(kg + (long)(kb + 1) * 32 * 64); if (pass == 2) stV = *(const uint4*)(vg + (kb + 1) * 32); }
            if (wave_on) {
                // S^T tiles: sacc[qt][kt]: lane (q = r16, g) reg r <-> key 16*kt + 4*r + g
                f32x4 sacc[2][2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) sacc[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    H8 k0, k1;
                    const half_t* kp = &ldsK[buf][(kt * 16 + kappa) * AT_KROW + g * 8];
                    k0.u = *(const uint4*)kp; k1.u = *(const uint4*)(kp + 32);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float kv = h2f(k0.h[e]);
                        sacc[0][kt] = MFMA16(kv, qf[0][e], sacc[0][kt]);
                        sacc[1][kt] = MFMA16(kv, qf[1][e], sacc[1][kt]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float kv = h2f(k1.h[e]);
                        sacc[0][kt] = MFMA16(kv, qf[0][8 + e], sacc[0][kt]);
                        sacc[1][kt] = MFMA16(kv, qf[1][8 + e], sacc[1][kt]);
                    }
                }
                // scale + mask
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int key = kb * 32 + kt * 16 + 4 * r + g;
                            float sv = sacc[qt][kt][r] * kq_scale;
                            sacc[qt][kt][r] = (key < n_ctx) ? sv : -INFINITY;
                        }
                if (dbg2 && pass == 0 && b == 0 && h == 0 && blockIdx.x == 0 && wave == 0) {
                    for (int qt = 0; qt < 2; ++qt) for (int kt = 0; kt < 2; ++kt) for (int r = 0; r < 4; ++r) dbg2[(long)(qt * 16 + r16) * Tpad + kb * 32 + kt * 16 + 4 * r + g] = sacc[qt][kt][r];
                }
                if (pass == 0) {
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) rmax[qt] = fmaxf(rmax[qt], sacc[qt][kt][r]);
                } else if (pass == 1) {
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) rsum[qt] += (double)skw_expf(sacc[qt][kt][r] - rmax[qt]);
                } else {
                    H8 vf[4];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) vf[ct].u = *(const uint4*)(&ldsV[buf][(ct * 16 + r16) * AT_VROW + g * 8]);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float p = skw_expf(sacc[qt][kt][r] - rmax[qt]) * rinv[qt];
                                p = h2f(f2h(p));
                                if (dbg2 && b == 0 && h == 0 && blockIdx.x == 0 && wave == 0) dbg2[(long)(32 + qt * 16 + r16) * Tpad + kb * 32 + kt * 16 + 4 * r + g] = p;
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct) oacc[qt][ct] = MFMA16(p, h2f(vf[ct].h[kt * 4 + r]), oacc[qt][ct]);
                            }
                }
            }
            if (kb + 1 < nkb) {
                *(uint4*)(&ldsK[buf ^ 1][krow * AT_KROW + kcc * 8]) = stK;
                if (pass == 2) *(uint4*)(&ldsV[buf ^ 1][vrow * AT_VROW + vcc * 8]) = stV;
            }
            __syncthreads();
        }
        if (pass == 0) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) { rmax[qt] = fmaxf(rmax[qt], __shfl_xor(rmax[qt], 16, 64)); rmax[qt] = fmaxf(rmax[qt], __shfl_xor(rmax[qt], 32, 64)); }
        } else if (pass == 1) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) { rsum[qt] += __shfl_xor(rsum[qt], 16, 64); rsum[qt] += __shfl_xor(rsum[qt], 32, 64); rinv[qt] = (float)(1.0 / rsum[qt]); }
        }
    }
    if (!wave_on) return;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int qi = q0 + qt * 16 + g * 4 + r; int c = ct * 16 + r16;
                if (qi < n_ctx) att_store(out, ((long)b * n_ctx + qi) * ld_out, h * 64 + c, oacc[qt][ct][r], f32_out);
                if (dbg && qi < n_ctx && b == 0) { dbg[(long)qi * (H * 64) + h * 64 + c] = oacc[qt][ct][r];
                if (ct == 0 && r == 0 && g == 0 && q0 + qt * 16 + r16 < n_ctx) { dbg[(long)n_ctx * H * 64 + (long)h * n_ctx + q0 + qt * 16 + r16] = rmax[qt];
                dbg[(long)n_ctx * H * 64 + (long)(H + h) * n_ctx + q0 + qt * 16 + r16] = rinv[qt]; } }
            }
}
// ------------------------------------------------------------------ encoder self-attention, single-pass form (K5)
// On gfx950 the f32 MFMA and the vector ALU do not overlap on a SIMD, not even across waves (tools/probe/probe_mfma_chain.hip: 16 conversions
// + 8 MFMAs cost 41.5 cycles/MFMA at 1, 2 or 3 waves per SIMD), so the kernel that wins is the one with the fewest MFMA + VALU cycles:
// Q.K^T once (scores stay on chip: RT tiles in registers/AGPRs, the rest in a per-wave LDS slab), one exponential per score,
// evaluated two at a time with packed f32 math, and every dependent MFMA chain issued back to back.  One wave = 16 queries against all
// keys; the four waves of a workgroup are independent (no barriers).  Arithmetic per output is identical to k_attn_encoder:
// d-ascending score chains, exact row max, skw_expf, f64 row sum, P = f16(e * inv), key-ascending P.V chains.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// skw_expf for a pair of non-positive arguments: same operations as include/skw_math.h (the x > 88.7 and n > 127 branches cannot
// trigger for x <= 0; y * 2^n == ldexp(y, n) exactly while the result is normal, which x >= -86 guarantees)
__device__ __forceinline__ f32x2 expf_nonpos_x2(f32x2 x) {
    f32x2 t = x * (f32x2){1.44269504088896341f, 1.44269504088896341f};
    f32x2 n = {__builtin_rintf(t[0]), __builtin_rintf(t[1])};
    f32x2 r = __builtin_elementwise_fma(n, (f32x2){-0.693359375f, -0.693359375f}, x);
    r = __builtin_elementwise_fma(n, (f32x2){2.12194440e-4f, 2.12194440e-4f}, r);
    f32x2 p = {1.9875691500e-4f, 1.9875691500e-4f};
    p = __builtin_elementwise_fma(p, r, (f32x2){1.3981999507e-3f, 1.3981999507e-3f});
    p = __builtin_elementwise_fma(p, r, (f32x2){8.3334519073e-3f, 8.3334519073e-3f});
    p = __builtin_elementwise_fma(p, r, (f32x2){4.1665795894e-2f, 4.1665795894e-2f});
    p = __builtin_elementwise_fma(p, r, (f32x2){1.6666665459e-1f, 1.6666665459e-1f});
    p = __builtin_elementwise_fma(p, r, (f32x2){5.0000001201e-1f, 5.0000001201e-1f});
    f32x2 r2 = r * r;
    f32x2 y = __builtin_elementwise_fma(p, r2, r) + (f32x2){1.0f, 1.0f};
    f32x2 o;
    o[0] = (x[0] >= -86.0f) ? __builtin_ldexpf(y[0], (int)n[0]) : 0.0f;
    o[1] = (x[1] >= -86.0f) ? __builtin_ldexpf(y[1], (int)n[1]) : 0.0f;
    return o;
}
// The same without the x >= -86 -> 0 clamp, for finite arguments inside a softmax: below -86 the unclamped value is some
// e < 2^-124 (or 0) instead of exactly 0, which neither the f64 row sum (>= 1: the row maximum contributes exp(0)) nor
// P = f16(e / sum) can see.  Not for -inf (masked keys): the reduction would produce NaN.
__device__ __forceinline__ f32x2 expf_nonpos_x2_softmax(f32x2 x) {
    f32x2 t = x * (f32x2){1.44269504088896341f, 1.44269504088896341f};
    f32x2 n = {__builtin_rintf(t[0]), __builtin_rintf(t[1])};
    f32x2 r = __builtin_elementwise_fma(n, (f32x2){-0.693359375f, -0.693359375f}, x);
    r = __builtin_elementwise_fma(n, (f32x2){2.12194440e-4f, 2.12194440e-4f}, r);
    f32x2 p = {1.9875691500e-4f, 1.9875691500e-4f};
    p = __builtin_elementwise_fma(p, r, (f32x2){1.3981999507e-3f, 1.3981999507e-3f});
    p = __builtin_elementwise_fma(p, r, (f32x2){8.3334519073e-3f, 8.3334519073e-3f});
    p = __builtin_elementwise_fma(p, r, (f32x2){4.1665795894e-2f, 4.1665795894e-2f});
    p = __builtin_elementwise_fma(p, r, (f32x2){1.6666665459e-1f, 1.6666665459e-1f});
    p = __builtin_elementwise_fma(p, r, (f32x2){5.0000001201e-1f, 5.0000001201e-1f});
    f32x2 r2 = r * r;
    f32x2 y = __builtin_elementwise_fma(p, r2, r) + (f32x2){1.0f, 1.0f};
    return (f32x2){__builtin_ldexpf(y[0], (int)n[0]), __builtin_ldexpf(y[1], (int)n[1])};
}
template <int NT, int RT>
__global__ __launch_bounds__(256, 1) void k_attn_encoder_v3(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out,
                                                            int H, int n_ctx, int Tpad, float kq_scale, int qtiles, int f32_out) {
    constexpr int LT = NT - RT, NB = NT / 2, KD = 3;
    static_assert(NT % 2 == 0 && RT % 2 == 0 && RT <= NT, "tiles come in 32-key blocks");
    extern __shared__ __attribute__((aligned(16))) char smem_att[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4* slds = (f32x4*)smem_att + (size_t)wave * (LT > 0 ? LT : 1) * 64 + lane;     // tile t >= RT of this lane at slds[(t - RT) * 64]
    const int nblk = gridDim.x; int bid = blockIdx.x;
    { int q = nblk >> 3, r = nblk & 7, x = bid & 7, y = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y; }   // XCD-aware: one (batch, head) per L2
    const long bh = bid / qtiles; const int qt = bid % qtiles;
    const int b = (int)(bh / H), h = (int)(bh % H);
    const int q0 = qt * 64 + wave * 16;
    if (q0 >= n_ctx) return;                       // waves are independent: no workgroup barriers below
    const int r16 = lane & 15, g = lane >> 4;
    float qf[16];                                  // Q fragment (B operand): lane (q = r16, g) holds Q[q][d = 32*blk + 4*e + g]
    {
        int qi = q0 + r16; if (qi >= n_ctx) qi = n_ctx - 1;
        const half_t* qp = Qh + (bh * Tpad + qi) * 64 + g * 8;
        H8v a, c; a.v = *(const u32x4*)qp; c.v = *(const u32x4*)(qp + 32);
#pragma unroll
        // the softmax scale 64^-1/2 = 2^-3 is folded into Q: scaling by a power of two commutes with every rounding of the fma chain
        // (no product of two f16 values comes near the f32 subnormal range), so chain(q/8, k) == chain(q, k) * 0.125f bit for bit
        for (int e = 0; e < 8; ++e) { qf[e] = h2f(a.h[e]) * kq_scale; qf[8 + e] = h2f(c.h[e]) * kq_scale; }
    }
    const int kappa = 4 * (r16 & 3) + (r16 >> 2);  // MFMA row rho of a 16-key tile holds key kappa(rho): keeps the P.V chain ascending
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + bh * Tpad * 64), 0, (unsigned)(Tpad * 64 * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + bh * 64 * Tpad), 0, (unsigned)(64 * Tpad * 2), 0x00020000);
    const unsigned ko = (unsigned)((kappa * 64 + g * 8) * 2);                 // + kb*4096 per 32-key block, +2048 second tile, +64 d-block 1
    const unsigned vo = (unsigned)((r16 * Tpad + g * 8) * 2);                 // + ct*16*Tpad*2, + kb*64 per block
    f32x4 sreg[RT > 0 ? RT : 1];
    float rmax = -INFINITY;
    H8v ring[KD][4];
    // ---- phase A: S^T tile by tile.  Per tile: 16 conversions, then its 16-MFMA chain back to back; the previous tile is
    //      scaled / masked / max'ed after the chain has been issued, so nothing waits on an MFMA result.
#pragma unroll
    for (int j = 0; j < KD; ++j) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ring[j][u].v = __builtin_amdgcn_raw_buffer_load_b128(rk, ko + j * 4096 + (u >> 1) * 2048 + (u & 1) * 64, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 prev = {0.f, 0.f, 0.f, 0.f};
    auto finish_tile = [&](int t, f32x4 a) {      // t = tile index of `a`
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = a[r];
            if (t == NT - 1) v = (t * 16 + 4 * r + g < n_ctx) ? v : -INFINITY;     // only the last tile can reach past n_ctx (host checks Tpad - n_ctx < 16)
            a[r] = v; rmax = fmaxf(rmax, v);
        }
        if (t < RT) sreg[t < RT ? t : 0] = a; else slds[(t - RT) * 64] = a;
    };
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        H8v c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = ring[kb % KD][u];
#pragma unroll
        for (int u = 0; u < 4; ++u) ring[kb % KD][u].v = __builtin_amdgcn_raw_buffer_load_b128(rk, ko + (kb + KD) * 4096 + (u >> 1) * 2048 + (u & 1) * 64, 0, 0);   // past Tpad: zeros
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float xk[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xk[e] = h2f(c[kt * 2].h[e]); xk[8 + e] = h2f(c[kt * 2 + 1].h[e]); }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; ++e) acc = MFMA16(xk[e], qf[e], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (kb * 2 + kt > 0) finish_tile(kb * 2 + kt - 1, prev);
            prev = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    finish_tile(NT - 1, prev);
    // V^T ring starts now so the loads fly during the exponentials
#pragma unroll
    for (int j = 0; j < KD; ++j) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) ring[j][ct].v = __builtin_amdgcn_raw_buffer_load_b128(rv, vo + ct * 16 * Tpad * 2 + j * 64, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 16, 64)); rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    // ---- phase B: e = expf(s - max) in place (pairs, packed math), f64 row sum
    double rsum = 0.0;
    const f32x2 mx = {rmax, rmax};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 v = (t < RT) ? sreg[t < RT ? t : 0] : slds[(t - RT) * 64];
        f32x2 e0, e1;
        if (t == NT - 1) { e0 = expf_nonpos_x2((f32x2){v[0], v[1]} - mx); e1 = expf_nonpos_x2((f32x2){v[2], v[3]} - mx); }      // may hold masked (-inf) keys
        else { e0 = expf_nonpos_x2_softmax((f32x2){v[0], v[1]} - mx); e1 = expf_nonpos_x2_softmax((f32x2){v[2], v[3]} - mx); }
        rsum += (double)e0[0]; rsum += (double)e0[1]; rsum += (double)e1[0]; rsum += (double)e1[1];
        v = (f32x4){e0[0], e0[1], e1[0], e1[1]};
        if (t < RT) sreg[t < RT ? t : 0] = v; else slds[(t - RT) * 64] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
    rsum += __shfl_xor(rsum, 16, 64); rsum += __shfl_xor(rsum, 32, 64);
    const float rinv = (float)(1.0 / rsum);
    // ---- phase C: O = P V with P = f16(e * inv) as the A operand, keys ascending; four channel tiles = four interleaved chains
    f32x4 oacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        H8v vf[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) vf[ct] = ring[kb % KD][ct];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) ring[kb % KD][ct].v = __builtin_amdgcn_raw_buffer_load_b128(rv, (kb + KD < NB) ? vo + ct * 16 * Tpad * 2 + (kb + KD) * 64 : 0x7fffff00u, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int t = kb * 2 + kt;
            f32x4 ev = (t < RT) ? sreg[t < RT ? t : 0] : slds[(t - RT) * 64];
            float pr[4], xv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = h2f(f2h(ev[r] * rinv));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) xv[ct][r] = h2f(vf[ct].h[kt * 4 + r]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) oacc[ct] = MFMA16(pr[r], xv[ct][r], oacc[ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int qi = q0 + g * 4 + r; int c = ct * 16 + r16;
            if (qi < n_ctx) att_store(out, ((long)b * n_ctx + qi) * ld_out, h * 64 + c, oacc[ct][r], f32_out);
        }
}

void skw_attn_encoder(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out, int B, int H, int n_ctx, int Tpad, hipStream_t s, float* dbg, float* dbg2, int f32_out) {
    // Whisper's 1500-frame context; k_attn_encoder below is the form that carries the stage taps (layer 0 of a tapped run) and any other context length
    if (Tpad == 1504 && Tpad - n_ctx < 16 && !dbg) {
        const int qtiles = (n_ctx + 63) / 64;
        constexpr int RT3 = 72;
        hipLaunchKernelGGL((k_attn_encoder_v3<94, RT3>), dim3(qtiles * H * B), dim3(256), (size_t)4 * (94 - RT3) * 64 * 16, s, Qh, Kh, Vt, out, ld_out, H, n_ctx, Tpad,
            1.0f / sqrtf(64.0f), qtiles, f32_out);
        return;
    }
    dim3 grid((n_ctx + 127) / 128, H, B);
    hipLaunchKernelGGL(k_attn_encoder, grid, dim3(256), 0, s, Qh, Kh, Vt, out, ld_out, H, n_ctx, Tpad, 1.0f / sqrtf(64.0f), dbg, dbg2, f32_out);
}

// ------------------------------------------------------------------ log-mel front end (K1)
// one 128-thread block per frame; restates whisper.cpp fft()/dft() operation by operation (f32, no contraction)
__global__ __launch_bounds__(128) void k_mel_frames(const float* pcm, const long* pcm_off, const int* n_samples, const int* n_len, int n_len_max,
                                                    SkwMelTables t, float* mel_raw) {
    __shared__ float x[400];
    __shared__ float bufA[800], bufB[800];
    __shared__ float cs_t[400], sn_t[400];           // twiddles: 25 x 400 + 4 x 200 reads per frame, served from LDS instead of L1/L2
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ns_raw = n_samples[b], nl = n_len[b];
    if (i >= nl) return;
    float* outp = mel_raw + ((long)b * n_len_max + i) * t.n_mel;
    const int ns = ns_raw + 200;
    int n_calc = ns / 160 + 1; if (n_calc > nl) n_calc = nl;
    if (i >= n_calc) { if (tid < t.n_mel) outp[tid] = (float)log10(1e-10); return; }
    for (int j = tid; j < 400; j += 128) { cs_t[j] = t.cos_t[j]; sn_t[j] = t.sin_t[j]; }
    const float* p = pcm + pcm_off[b];
    const int offset = i * 160;
    int lim = ns - offset; if (lim > 400) lim = 400;
    for (int j = tid; j < 400; j += 128) {
        float v = 0.0f;
        if (j < lim) {
            int pp = offset + j; float sm;
            if (pp < 200) { int q = 200 - pp; sm = (q < ns_raw) ? p[q] : 0.0f; }
            else { int q = pp - 200; sm = (q < ns_raw) ? p[q] : 0.0f; }
            v = t.hann[j] * sm;
        }
        x[j] = v;
    }
    __syncthreads();
    // 16 leaf DFTs of length 25: leaf o holds x[o + 16 n].  A thread owns one output bin k of up to four leaves (o = og, og + 5, og + 10, og + 15):
    // the twiddle of term n is the same for all of them, so it is fetched (and its index advanced) once per term instead of once per output —
    // each output is still its own n-ascending chain of separately rounded products and sums.
    if (tid < 125) {
        const int k = tid % 25, og = tid / 25;
        float re[4] = {0.0f, 0.0f, 0.0f, 0.0f}, im[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int step = (k * 16) % 400; int idx = 0;          // (k * n * 16) % 400 without a division per term
        const bool four = og == 0;                             // leaf 15 exists only for og = 0
        for (int n = 0; n < 25; ++n) {
            const float c = cs_t[idx], sn = sn_t[idx];
#pragma unroll
            for (int j = 0; j < 3; ++j) { const float xin = x[og + 5 * j + 16 * n]; re[j] += xin * c; im[j] -= xin * sn; }
            if (four) { const float xin = x[15 + 16 * n]; re[3] += xin * c; im[3] -= xin * sn; }
            idx += step; if (idx >= 400) idx -= 400;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int o = og + 5 * j; bufA[2 * (o * 25 + k)] = re[j]; bufA[2 * (o * 25 + k) + 1] = im[j]; }
        if (four) { bufA[2 * (15 * 25 + k)] = re[3]; bufA[2 * (15 * 25 + k) + 1] = im[3]; }
    }
    __syncthreads();
    float* src = bufA; float* dst = bufB;
    for (int N = 50; N <= 400; N <<= 1) {
        const int half_n = N >> 1, groups = 400 / N, step = 400 / N;
        for (int u = tid; u < 200; u += 128) {
            int gi = u / half_n, k = u % half_n;
            int idx = k * step;
            float re = cs_t[idx], im = -sn_t[idx];
            const float* ev = src + 2 * (gi * half_n + k); const float* od = src + 2 * ((gi + groups) * half_n + k);
            float er = ev[0], ei = ev[1], orr = od[0], oi = od[1];
            float* o0 = dst + 2 * (gi * N + k); float* o1 = dst + 2 * (gi * N + k + half_n);
            o0[0] = (er + re * orr) - im * oi;
            o0[1] = (ei + re * oi) + im * orr;
            o1[0] = (er - re * orr) + im * oi;
            o1[1] = (ei - re * oi) - im * orr;
        }
        __syncthreads();
        float* tmp = src; src = dst; dst = tmp;
    }
    // power spectrum into x[0..200]
    for (int j = tid; j < 201; j += 128) { float re = src[2 * j], im = src[2 * j + 1]; x[j] = re * re + im * im; }
    __syncthreads();
    if (tid < t.n_mel) {
        const float* fl = t.filters + (long)tid * t.n_fft_bins;
        // whisper.cpp's loop runs over all bins in groups of four; a group whose four taps are all zero adds +0.0 to the f64 sum, so
        // only the groups [g_lo, g_hi) that hold this filter's non-zero taps are visited (same groups, same order, same bits)
        double sum = 0.0; int k = t.grp_lo ? 4 * t.grp_lo[tid] : 0;
        const int k_end = t.grp_hi ? min(4 * t.grp_hi[tid], t.n_fft_bins - 3) : t.n_fft_bins - 3;
        for (; k < k_end; k += 4) {
            float s4 = x[k] * fl[k] + x[k + 1] * fl[k + 1] + x[k + 2] * fl[k + 2] + x[k + 3] * fl[k + 3];
            sum += (double)s4;
        }
        for (k = (t.n_fft_bins - 3 > 0) ? ((t.n_fft_bins - 3 + 3) / 4) * 4 : 0; k < t.n_fft_bins; ++k) sum += (double)(x[k] * fl[k]);
        sum = log10(sum > 1e-10 ? sum : 1e-10);
        outp[tid] = (float)sum;
    }
}
void skw_mel_frames(const float* pcm, const long* pcm_off, const int* n_samples, const int* n_len, int B, int n_len_max, SkwMelTables t, float* mel_raw, hipStream_t s) {
    hipLaunchKernelGGL(k_mel_frames, dim3(n_len_max, B), dim3(128), 0, s, pcm, pcm_off, n_samples, n_len, n_len_max, t, mel_raw);
}

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__global__ void k_mel_max(const float* mel, const int* n_len, int n_len_max, int n_mel, unsigned* clip_max_ord) {
    const int b = blockIdx.y; const long n = (long)n_len[b] * n_mel; const float* p = mel + (long)b * n_len_max * n_mel;
    float m = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, p[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&clip_max_ord[b], f2ord(m));
}
__global__ void k_mel_norm(float* mel, const int* n_len, int n_len_max, int n_mel, const unsigned* clip_max_ord) {
    const int b = blockIdx.y; const long n = (long)n_len[b] * n_mel; float* p = mel + (long)b * n_len_max * n_mel;
    double mmax = (double)ord2f(clip_max_ord[b]) - 8.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = p[i]; if ((double)v < mmax) v = (float)mmax;
        p[i] = (float)(((double)v + 4.0) / 4.0);
    }
}
void skw_mel_normalize(float* mel, const int* n_len, int B, int n_len_max, int n_mel, float* clip_max, hipStream_t s) {
    hipMemsetAsync(clip_max, 0, sizeof(unsigned) * B, s);
    hipLaunchKernelGGL(k_mel_max, dim3(64, B), dim3(256), 0, s, mel, n_len, n_len_max, n_mel, (unsigned*)clip_max);
    hipLaunchKernelGGL(k_mel_norm, dim3(64, B), dim3(256), 0, s, mel, n_len, n_len_max, n_mel, (const unsigned*)clip_max);
}
// conv1 im2col: out[(bw*T + t)][kperm(k)], k = tap*n_mel + c (k < 3*n_mel), zero padded to k_pad (256 for 80 bands, 384 for large-v3's 128: the width of the conv1 weight image)
__global__ void k_mel_im2col(const float* mel, const int* clip_idx, const int* seek, const int* n_len, int n_len_max, int n_mel, int T, int k_pad, half_t* out) {
    const int bw = blockIdx.y, t = blockIdx.x;   // 256 threads
    const int clip = clip_idx[bw], sk = seek[bw], nl = n_len[clip];
    for (int k = threadIdx.x; k < k_pad; k += blockDim.x) {
        float v = 0.0f;
        if (k < 3 * n_mel) {
            int tap = k / n_mel, c = k % n_mel; int tt = t - 1 + tap; int fr = sk + tt;
            if (tt >= 0 && tt < T && fr < nl) v = mel[((long)clip * n_len_max + fr) * n_mel + c];
        }
        out[((long)bw * T + t) * k_pad + skw_kperm(k)] = f2h(v);
    }
}
void skw_mel_im2col(const float* mel, const int* clip_idx, const int* seek, const int* n_len, int Bw, int n_len_max, int n_mel, int T, int k_pad, half_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_mel_im2col, dim3(T, Bw), dim3(256), 0, s, mel, clip_idx, seek, n_len, n_len_max, n_mel, T, k_pad, out);
}

// ------------------------------------------------------------------ decoder pieces
__global__ void k_dec_embed(const half_t* te, const float* pe, const int* tok, const int* pos, int d, float* x) {
    const int b = blockIdx.x; const int tk = tok[b * (int)(sizeof(SkwSeqState) / 4)]; const int ps = pos[b * (int)(sizeof(SkwSeqState) / 4)];
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(long)b * d + i] = h2f(te[(long)tk * d + skw_kperm(i)]) + pe[(long)ps * d + i];
}
// the same with the first layer's LayerNorm attached: one wave per row, x written in passing (d <= 1536)
template <int NC, bool FULL>
__global__ __launch_bounds__(256) void k_dec_embed_ln(const half_t* te, const float* pe, const int* tok, const int* pos, int B, int d, float* x, const float* w, const float* b, half_t* y16) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int tk = tok[row * (int)(sizeof(SkwSeqState) / 4)], ps = pos[row * (int)(sizeof(SkwSeqState) / 4)];
    float v[1][NC], wv[NC], bv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = lane + 64 * c; const bool in = FULL || i < d;
        v[0][c] = in ? h2f(te[(long)tk * d + skw_kperm(i)]) + pe[(long)ps * d + i] : 0.0f; wv[c] = in ? w[i] : 0.0f; bv[c] = in ? b[i] : 0.0f;
        if (in) x[(long)row * d + i] = v[0][c];
    }
    const bool live[1] = {true}; half_t* const o16[1] = {y16 + (long)row * d}; float* const o32[1] = {nullptr};
    skw_ln_rows<1, NC, FULL>(v, wv, bv, d, lane, live, o16, o32);
}
void skw_dec_embed(const half_t* te, const float* pe, const int* tok, const int* pos, int B, int d, float* x, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_embed, dim3(B), dim3(256), 0, s, te, pe, tok, pos, d, x);
}
void skw_dec_embed_ln(const half_t* te, const float* pe, const int* tok, const int* pos, int B, int d, float* x, const float* w, const float* b, half_t* y16, hipStream_t s) {
    if (d == 768) hipLaunchKernelGGL((k_dec_embed_ln<12, true>), dim3((B + 3) / 4), dim3(256), 0, s, te, pe, tok, pos, B, d, x, w, b, y16);
    else if (d <= 768) hipLaunchKernelGGL((k_dec_embed_ln<12, false>), dim3((B + 3) / 4), dim3(256), 0, s, te, pe, tok, pos, B, d, x, w, b, y16);
    else hipLaunchKernelGGL((k_dec_embed_ln<24, false>), dim3((B + 3) / 4), dim3(256), 0, s, te, pe, tok, pos, B, d, x, w, b, y16);
}

__device__ __forceinline__ float block_max(float v, float* sh) {
    v = skw_wave_max_f32(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0]; for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
    v = wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0; for (int i = 0; i < nw; ++i) r += sh[i];
    __syncthreads();
    return r;
}

// single-query attention over n_kv keys with plain f16 K/V rows (row stride ldkv halves), head dim 64.
//   s_j = chain_d q[d] K[j][d];  softmax as ggml_soft_max_ext (scale 1, f64 denominator);  out[c] = chain_j f16(p_j) V[j][c]
// One wave per (sequence, head), no block-level synchronisation: lane = key for the scores (64 independent d-chains per
// pass over K), lane = channel for P.V (one key-ascending chain per output, p_j broadcast with v_readlane).
// A block is 4 waves = 4 heads of one sequence.
template <int MAXT, bool FASTV = false>      // FASTV (f16_mfma only): the V rows arrive as 16-byte pieces and each lane sums its own keys' share — see below
__global__ __launch_bounds__(256) void k_dec_attn(const half_t* q, long ldq, const half_t* kbase, const half_t* vbase, long batch_stride, long ldkv,
                                                  const int* n_kv_ptr, int n_kv_stride, int n_kv_fixed, int H, half_t* out, long ldo, const int* active, int active_stride, int f32_out, SkwQ8Out q8,
                                                  const int* seq, int ofrag_k) {
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (h >= H) return;
    const int act = active ? active[b * active_stride] : 1;   // (tested after the prefetches below are on their way: one round trip for all of them)
    const int bs = seq ? seq[b * active_stride] : b;          // the prompt pass: row b is one prompt token of sequence seq[b] (one more round trip there; the step has seq == null)
    const half_t* K = kbase + (long)bs * batch_stride + h * 64;
    const half_t* V = vbase + (long)bs * batch_stride + h * 64;
    // Everything whose address depends on nothing loaded is requested before the first wait, in the order it is needed (a wave's loads return in
    // order): q and the first 64 K rows, which the score chains start on, then the V rows of the first NPRE keys, which arrive under them (rows
    // past n_kv exist in the cache and are simply not used): the short-context steps pay one round trip, not one per 16 keys after the softmax.
    constexpr int NPRE = (MAXT * 64 < 128) ? MAXT * 64 : 128;
    const int rows_cap = (int)(batch_stride / ldkv);
    uint4 qraw[8];
    { const uint4* qp = (const uint4*)(q + (long)b * ldq + h * 64);
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) qraw[c8] = qp[c8]; }
    uint4 kk0[8];
    { const uint4* kr = (const uint4*)(K + (long)min(lane, rows_cap - 1) * ldkv);
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) kk0[c8] = kr[c8]; }
    __builtin_amdgcn_sched_barrier(0);
    // FASTV.  The exact form below gives lane c the key-ascending fma chain of channel c: one 2-byte load per key and lane (128 load instructions for the prefetched keys, about a
    // microsecond of issue in a 8.6 us kernel) and a readlane + convert + fma per key.  The tolerance precision owes no order: lane (kg = lane >> 3, piece = lane & 7) takes the
    // 16-byte piece `piece` of the rows of keys kg, kg + 8, ... (one load instruction = eight whole 128-byte head rows), sums p_key * v over ITS keys for its eight channels, and the
    // eight key groups meet in three shuffle steps.  16 load instructions instead of 128, 8 fmas per key and lane group instead of 64.
    constexpr int NPV = FASTV ? NPRE / 8 : 1;
    half_t vpre[FASTV ? 1 : NPRE];
    uint4 vq[NPV];
    const int kg = lane >> 3, piece = lane & 7;
    if constexpr (FASTV) {
#pragma unroll
        for (int u = 0; u < NPV; ++u) vq[u] = *(const uint4*)(V + (long)min(u * 8 + kg, rows_cap - 1) * ldkv + piece * 8);
    } else {
        const half_t* vp0 = V + lane;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) vpre[u] = vp0[(long)min(u, rows_cap - 1) * ldkv];
    }
    __builtin_amdgcn_sched_barrier(0);
    float qv[64];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) { H8 t; t.u = qraw[c8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[c8 * 8 + e] = h2f(t.h[e]); }
    const int n_kv = n_kv_ptr ? (n_kv_ptr[b * n_kv_stride] + 1) : n_kv_fixed;
    if (!act) return;                                        // a finished sequence keeps its slot in the batch; its row is never sampled again
    float sc[MAXT];
    float lmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        sc[t] = -INFINITY;
        if (t * 64 < n_kv) {   // wave-uniform
            const int key = t * 64 + lane;
            uint4 kk[8];
            if (t == 0) {
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) kk[c8] = kk0[c8];
            } else {
                const uint4* kr = (const uint4*)(K + (long)min(key, n_kv - 1) * ldkv);
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) kk[c8] = kr[c8];
            }
            float a = 0.0f;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) { H8 t8; t8.u = kk[c8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a = __builtin_fmaf(qv[c8 * 8 + e], h2f(t8.h[e]), a); }
            if (key < n_kv) { sc[t] = a; lmax = fmaxf(lmax, a); }
        }
    }
    lmax = skw_wave_max_f32(lmax);
    double lsum = 0.0;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) if (t * 64 < n_kv) { float e = skw_expf(sc[t] - lmax); sc[t] = e; lsum += (double)e; }
    lsum = wave_sum_f64(lsum);
    const float inv = (float)(1.0 / lsum);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) if (t * 64 < n_kv) sc[t] = h2f(f2h(sc[t] * inv));
    if constexpr (FASTV) {
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = 0.0f;
        const int ngrp = (n_kv + 7) >> 3;                     // groups of eight keys (wave-uniform)
#pragma unroll
        for (int u = 0; u < NPV; ++u) {
            if (u < ngrp) {
                const int key = u * 8 + kg;
                const float pk = __shfl(sc[(u * 8) >> 6], key & 63);          // (the group's eight keys lie in one 64-key pass)
                if (key < n_kv) { H8 v8; v8.u = vq[u];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a8[e] = __builtin_fmaf(pk, h2f(v8.h[e]), a8[e]); }
            }
        }
        for (int u = NPV; u < ngrp; ++u) {                    // keys past the prefetched 128 (long prompts): one load per group
            const int key = u * 8 + kg;
            H8 v8; v8.u = *(const uint4*)(V + (long)min(key, n_kv - 1) * ldkv + piece * 8);
            float pk = 0.0f;
#pragma unroll
            for (int t = NPRE / 64; t < MAXT; ++t) if ((u * 8) >> 6 == t) pk = __shfl(sc[t], key & 63);
            if (key < n_kv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = __builtin_fmaf(pk, h2f(v8.h[e]), a8[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { a8[e] += __shfl_xor(a8[e], 8); a8[e] += __shfl_xor(a8[e], 16); a8[e] += __shfl_xor(a8[e], 32); }
        if (kg == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) att_store_m(out, b, ldo, h * 64 + piece * 8 + e, a8[e], f32_out, ofrag_k);
        }
        return;
    }
    // P.V: lane = channel
    float acc = 0.0f;
    const half_t* vp = V + lane;
#pragma unroll
    for (int g16 = 0; g16 < NPRE / 16; ++g16) {          // keys [0, NPRE) from the prefetched rows, ascending
        const int pbits = __float_as_int(sc[g16 >> 2]);
        if (g16 * 16 + 16 <= n_kv) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(pbits, (g16 & 3) * 16 + u)), h2f(vpre[g16 * 16 + u]), acc);
        } else if (g16 * 16 < n_kv) {
#pragma unroll
            for (int u = 0; u < 16; ++u) if (g16 * 16 + u < n_kv) acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(pbits, (g16 & 3) * 16 + u)), h2f(vpre[g16 * 16 + u]), acc);
        }
    }
#pragma unroll
    for (int t = NPRE / 64; t < MAXT; ++t) {
        if (t * 64 < n_kv) {
            const int nj = min(64, n_kv - t * 64);
            const int pbits = __float_as_int(sc[t]);
            const half_t* vt = vp + (long)t * 64 * ldkv;
            int jj = 0;
            for (; jj + 16 <= nj; jj += 16) {
                half_t vv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) vv[u] = vt[(long)(jj + u) * ldkv];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(pbits, jj + u)), h2f(vv[u]), acc);
            }
            for (; jj < nj; ++jj) acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(pbits, jj)), h2f(vt[(long)jj * ldkv]), acc);
        }
    }
    if (q8.q) {   // the projection that follows multiplies by block-quantised weights: the row leaves as ggml's q8 blocks (a head's 64 channels = two blocks)
        float dd, ss; const int qi = q8_half_wave_block(acc, &dd, &ss);
        const int i = h * 64 + lane, blk = i >> 5;
        q8.q[(long)b * ldo + i] = (int8_t)qi;
        if ((lane & 31) == 0) { q8.dT[(long)blk * q8.M + b] = dd; q8.sT[(long)blk * q8.M + b] = ss; }
        return;
    }
    att_store_m(out, b, ldo, h * 64 + lane, acc, f32_out, ofrag_k);
}
// ------------------------------------------------------------------ decoder cross-attention (K9), bandwidth form
// Two waves per (sequence, head), three heads per workgroup (12 heads x 64 sequences = 256 workgroups of 6 waves: every CU, 1.5 waves/SIMD).
// Scores: the pair splits the keys; lane = key, 64 independent d-ascending chains per pass.  K rows [key][768] plain are fetched
// coalesced (8 lanes x 16 B = one 128-byte head row, so each cache line is consumed by the instruction that fetched it) through a
// 3-deep register ring and transposed in a per-wave LDS slab (row stride 144 B: conflict-free ds_read_b128).
// P.V on the matrix cores, the pair splitting the channels: A = V^T fragment (cross V is stored per head as [channel][Tpad kperm], one
// 16-byte load = 8 keys of one channel), B = p broadcast to every column, so each output channel is the same key-ascending fma chain
// as the scalar form; an 8-deep ring keeps V^T in flight (4 .. 16 deep measure the same within 2 %).  HBM-bound: 55.3 MB per sequence per step over all layers.
// (HIP's uint4 arrays defeat SROA and land in scratch; the rings use ext_vector types.)
#ifndef SKW_XATTN_RD16
#define SKW_XATTN_RD16 16      // V^T blocks in flight per wave in the f16_mfma P.V (8: 56.4 us per launch in tools/xattn_probe.py, 12: 55.7, 16: 55.7, 24: 56.1)
#endif
template <int MAXT, int WPH, int HPW, bool PV16 = false>
__global__ __launch_bounds__(64 * HPW * WPH, (HPW * WPH >= 12) ? 1 : 12 / (HPW * WPH)) void k_dec_cross_attn(const half_t* q, long ldq, const half_t* kbase, long k_batch_stride, long ldk,
                                                           const half_t* vtbase, int n_ctx, int Tpad, int H, half_t* out, long ldo, const int* active, int active_stride,
                                                               int f32_out, const int* seq) {
    const int bx = blockIdx.x, by = blockIdx.y;
    if (active && !active[by * active_stride]) return;      // uniform per workgroup (one sequence): a finished sequence stops streaming its 55 MB of cross K/V
    __shared__ float plds[HPW][MAXT * 64];
    __shared__ __attribute__((aligned(16))) half_t klds[HPW * WPH][64 * 72];
    __shared__ float smax[HPW][WPH];
    __shared__ double ssum[HPW][WPH];
    __shared__ __attribute__((aligned(16))) half_t p16[PV16 ? HPW : 1][PV16 ? MAXT * 64 : 8];      // PV16: the probabilities as f16 in kperm order, the f16 MFMA's second operand
    // half = this wave's part of the head (keys in the score phase, channels in P.V); readfirstlane: keeps the buffer descriptor in SGPRs (no waterfall loops)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hs = w / WPH, half = w % WPH;
    const int hraw = bx * HPW + hs, b = by;
    const bool valid = hraw < H;               // no early return: the pair meets at workgroup barriers
    const int h = valid ? hraw : H - 1;
    const int bs = seq ? seq[b * active_stride] : b;          // the prompt pass: row b attends over the cross K / V of sequence seq[b]
    const half_t* K = kbase + (long)bs * k_batch_stride + h * 64;
    auto ldk16 = [&](const half_t* p) -> u32x4 { return *(const u32x4*)p; };   // (non-temporal loads of the once-read K / V^T bytes measured 2 us slower per launch: 62.7 vs 60.4)
    H8v qh[8];                                  // q stays packed (32 VGPRs instead of 64); converted in the chain's shadow
    const int nt = (n_ctx + 63) >> 6, nth = (nt + WPH - 1) / WPH;
    const int t_lo = half * nth, t_hi = min(nt, t_lo + nth);
    float lmax = -INFINITY;
    half_t* kl = klds[w];
    float* pl = plds[hs];
    const int lrow = lane >> 3, lseg = lane & 7;
    u32x4 k0[8], k1[8], k2[8];
    auto kfill = [&](u32x4 (&dst)[8], int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = ldk16(K + (long)min(t * 64 + i * 8 + lrow, n_ctx - 1) * ldk + lseg * 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        const u32x4* qp = (const u32x4*)(q + (long)b * ldq + h * 64);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) qh[c8].v = qp[c8];
        kfill(k0, t_lo); kfill(k1, t_lo + 1);
    }
    // one pass = 64 keys: refill the free ring slot with pass t+2 first, then consume `cur`.  Roles rotate by name (three passes per
    // loop trip) so no register copies force early waits, and the scheduling barriers keep the loads where they are written.
    auto pass = [&](int t, u32x4 (&cur)[8], u32x4 (&fill)[8]) {
        // (wave-uniform) no refill past this wave's key range: the two passes after it belong to the next wave of the head, which fetches them itself — unconditional
        // refills were a third more K requests per wave (8 passes fetched for 6 consumed), and the PMC pass showed them arriving from memory, not from a cache:
        // 338 MB read per 64-row launch against 295 MB algorithmic
        if (t + 2 < t_hi) {
#pragma unroll
            for (int i = 0; i < 8; ++i) fill[i] = ldk16(K + (long)min((t + 2) * 64 + i * 8 + lrow, n_ctx - 1) * ldk + lseg * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t < t_hi) {       // wave-uniform
            const int key = t * 64 + lane;
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) *(u32x4*)(kl + (i * 8 + lrow) * 72 + lseg * 8) = cur[i];
            __builtin_amdgcn_wave_barrier();   // same wave, LDS in order: a compiler-level fence is all that is needed
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) asm volatile("" : "+v"(qh[c8].v));   // opaque: stops the 64 conversions being hoisted out of the loop into 64 live registers
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) { H8v t8; t8.v = *(const u32x4*)(kl + lane * 72 + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = __builtin_fmaf(h2f(qh[c8].h[e]), h2f(t8.h[e]), a); }
            __builtin_amdgcn_wave_barrier();
            if (key >= n_ctx) a = -INFINITY;
            pl[key] = a; lmax = fmaxf(lmax, a);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = t_lo; t < t_hi; t += 3) { pass(t, k0, k2); pass(t + 1, k1, k0); pass(t + 2, k2, k1); }
    // V^T ring: start the first loads before the softmax so they fly during it
    const int r16 = lane & 15, g = lane >> 4;
    const long bh = (long)bs * H + h;
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(vtbase + bh * 64 * Tpad), 0, (unsigned)(64 * Tpad * 2), 0x00020000);
    constexpr int CT = 4 / WPH;                                                   // channel tiles per wave
    const unsigned vo = (unsigned)(((half * CT * 16 + r16) * Tpad + g * 8) * 2);
    const int nkb = Tpad >> 5;
    constexpr int RD = PV16 ? SKW_XATTN_RD16 : 8;
    H8v ring[RD][CT];
#pragma unroll
    for (int j = 0; j < RD; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { ring[j][ct].v = __builtin_amdgcn_raw_buffer_load_b128(rv, (j < nkb) ? vo + ct * 16 * Tpad * 2 + j * 64 : 0x7fffff00u, 0, 0);
        __builtin_amdgcn_sched_barrier(0); }
    lmax = skw_wave_max_f32(lmax);
    if (lane == 0) smax[hs][half] = lmax;
    __syncthreads();
    lmax = smax[hs][0];
#pragma unroll
    for (int i = 1; i < WPH; ++i) lmax = fmaxf(lmax, smax[hs][i]);
    double lsum = 0.0;
    for (int t = t_lo; t < t_hi; ++t) { float e = skw_expf(pl[t * 64 + lane] - lmax); pl[t * 64 + lane] = e; lsum += (double)e; }   // exp(-inf) == 0 for masked keys
    lsum = wave_sum_f64(lsum);
    if (lane == 0) ssum[hs][half] = lsum;
    __syncthreads();
    double tot = ssum[hs][0];
#pragma unroll
    for (int i = 1; i < WPH; ++i) tot += ssum[hs][i];
    const float inv = (float)(1.0 / tot);
    if constexpr (PV16) {
        for (int t = t_lo; t < t_hi; ++t) p16[hs][skw_kperm(t * 64 + lane)] = f2h(pl[t * 64 + lane] * inv);
        for (int t = nt + half; t < MAXT; t += WPH) p16[hs][t * 64 + lane] = (half_t)0.0f;
    } else {
    for (int t = t_lo; t < t_hi; ++t) pl[t * 64 + lane] = h2f(f2h(pl[t * 64 + lane] * inv));
    for (int t = nt + half; t < MAXT; t += WPH) pl[t * 64 + lane] = 0.0f;      // key slots past the last pass (V^T pad is zero as well)
    }
    __syncthreads();
    f32x4 oacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb0 = 0; kb0 < nkb; kb0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int kb = kb0 + j;
            H8v vf[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) vf[ct] = ring[j][ct];
            const int nb = kb + RD;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) ring[j][ct].v = __builtin_amdgcn_raw_buffer_load_b128(rv, (nb < nkb) ? vo + ct * 16 * Tpad * 2 + nb * 64 : 0x7fffff00u, 0, 0);
            const int kbc = min(kb, nkb - 1);     // blocks past the end multiply p by zero-filled fragments
            if constexpr (PV16) {
                // f16_mfma precision: p is f16-valued and V^T is f16, so a 32-key block is ONE v_mfma_f32_16x16x32_f16 (16 cycles) instead of eight dependent
                // f32 MFMAs (256 cycles: tools/xattn_probe.py put them at 8.7 us of a 65 us launch that is otherwise at the memory system's pace).
                // Same products, summed in the matrix core's order for the block rather than key by key.
                typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
                const f16x8_t pb = *(const f16x8_t*)(&p16[hs][kbc * 32 + g * 8]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, vf[ct].v), pb, oacc[ct], 0, 0, 0);
                continue;
            }
            float pe[8], xv[CT][8];      // operands first, then the MFMAs back to back (a dependent MFMA directly behind its producer is the cheap case)
#pragma unroll
            for (int e = 0; e < 8; ++e) { pe[e] = pl[kbc * 32 + 4 * e + g];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xv[ct][e] = h2f(vf[ct].h[e]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) oacc[ct] = MFMA16(xv[ct][e], pe[e], oacc[ct]);   // O^T[c][*] += V^T[c][key] * p[key]
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D = O^T replicated over columns: lane (col n = r16, rows 4*g + r) -> channel 32*half + ct*16 + 4*g + r; take column 0
    if (r16 == 0 && valid) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) att_store(out, (long)b * ldo, h * 64 + (half * CT + ct) * 16 + 4 * g + r, oacc[ct][r], f32_out);
    }
}
// ------------------------------------------------------------------ decoder cross-attention (K9), f16_mfma precision: one streaming pass
// The tolerance precision owes the oracle no summation order, so the two-phase form above (all scores, a workgroup-wide softmax, then P.V — a barrier with the HBM
// stream drained in the middle of a 50 us launch) becomes one pass per wave with a running maximum.  Four waves per (row, head), each a contiguous quarter of the
// 32-key blocks; per block a wave takes 4 KB of K rows and 4 KB of V^T straight from memory into MFMA operands — no LDS, no transposes, no conversions:
//   scores   S[key][*] = K[key][:] . q on the f16 matrix cores: A = 16 K rows x 32 d (one 16-byte load per lane), B = q broadcast to every column; two d halves chain;
//            A row i of tile j is key 16 j + 4 (i & 3) + (i >> 2), so that lane group g ends up holding keys 4 e + g (e = 4 j + r) — the order V^T's 32-key blocks
//            are stored in (skw_kperm): the probabilities go into the next MFMA as its B operand as they stand
//   softmax  block maximum across the wave's four lane groups (two DPP-free shuffles per 8 KB), rescale of the 16 accumulators only when the maximum grows,
//            p = exp2((s - m) log2 e) rounded to f16 unnormalised (<= 1: no subnormal loss that 1 / sum would add), per-lane partial sums
//   P.V      O^T[c][*] += V^T[c][keys] . p, four 16-channel tiles, one MFMA each
// and the four partial (m, l, O) meet in LDS at the end (one barrier).  RD blocks (8 KB each) are in flight per wave, 12 waves per CU.
// K and V^T are the fragment-order images (skw_kfrag_off / skw_vtfrag_off): every load instruction is one contiguous KiB, and the loads carry the non-temporal policy (aux = 2: the
// once-read stream stays out of the caches the step's weights live in; 49.5 -> 45.7-46.2 us per full launch, profiles/r03h).  The four waves of a head take every fourth 32-key block, so
// together they walk one sequential stream.  What lost against this form (profiles/r03g, r03h, r04i): the row layouts (16 x 64 B per load instruction), contiguous quarters per wave, 2 / 4
// blocks in flight, one or two heads per workgroup, XCD-aware placement of a sequence's workgroups.
// clk (bench.py's roofline line; null in every other run): the launch's own begin and end on the device's constant-rate clock — see SkwKClk in skw_kernels.h.
template <int HPW, int RD>
__global__ __launch_bounds__(256 * HPW, (HPW >= 3) ? 1 : 3 / HPW) void k_dec_cross_attn16(const half_t* q, long ldq, const half_t* kbase, long k_batch_stride, long ldk, const half_t* vtbase,
                                                                                         int n_ctx, int Tpad, int H, half_t* out, long ldo, const int* active,
                                                                                             int active_stride, int f32_out, const int* seq, int ofrag_k, SkwKClk* clk) {
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    constexpr int AUX = 2;
    const int b = blockIdx.y;
    const bool row_live = !(active && !active[b * active_stride]);
    SkwKClkRec* rec = nullptr; unsigned long long t_in = 0;             // thread 0 only
    if (clk && threadIdx.x == 0) {
        // this workgroup's own launch count (finished rows count too): launches of one graph node are serial on its stream, the cell is nobody else's
        t_in = wall_clock64();
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < SKW_KCLK_MAX_WG) {
            const unsigned launch = clk->cnt[wg]; clk->cnt[wg] = launch + 1;
            if (row_live && launch < clk->cap) rec = &clk->rec[launch][wg % SKW_KCLK_SHARDS];
        }
    }
    if (!row_live) return;
    __shared__ float cmb[HPW][4][66];                                   // per wave: m, l, O[64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hs = w >> 2, part = w & 3;
    const int hraw = blockIdx.x * HPW + hs;
    const bool valid = hraw < H;
    const int h = valid ? hraw : H - 1;
    const int bs = seq ? seq[b * active_stride] : b;
    const int r16 = lane & 15, g = lane >> 4;
    // a wave's blocks: every fourth 32-key block (the four waves of a head walk one sequential stream together)
    const int nkb = Tpad >> 5;
    const int first = part, step = 4, cnt = (nkb - part + 3) >> 2;
    f16x8_t qb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qb[kk] = *(const f16x8_t*)(q + (long)b * ldq + h * 64 + kk * 32 + g * 8);
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(kbase + (long)bs * k_batch_stride), 0, (unsigned)((long)Tpad * ldk * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(vtbase + ((long)bs * H + h) * 64 * Tpad), 0, (unsigned)(64 * Tpad * 2), 0x00020000);
    u32x4 ring[RD][8];
    // one block's loads: 4 KiB of K (two score tiles x two d halves) and 4 KiB of V^T (four channel tiles); pad keys of the last tile hold whatever memory held (masked below)
    auto issue = [&](u32x4 (&slot)[8], int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) slot[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)(((h * (nkb * 2) + kb * 2) * 2 + i) * 1024 + lane * 16), 0, AUX);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) slot[4 + ct] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)((kb * 4 + ct) * 1024 + lane * 16), 0, AUX);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int j = 0; j < RD; ++j) if (j < cnt) issue(ring[j], first + j * step);
    constexpr float LOG2E = 1.44269504088896340736f;
    float m = -INFINITY, l = 0.0f;
    f32x4 o[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < cnt; n0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int n = n0 + j, kb = first + n * step;
            if (n < cnt) {                                              // wave-uniform
                f32x4 sc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ring[j][t * 2]), qb[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ring[j][t * 2 + 1]), qb[1], sc[t], 0, 0, 0);
                }
                if (kb * 32 + 32 > n_ctx) {                             // the block with the pad keys (their K rows were read from the last real key)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (kb * 32 + 4 * (4 * t + r) + g >= n_ctx) sc[t][r] = -INFINITY;
                }
                float bm = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
                bm = fmaxf(bm, __shfl_xor(bm, 16)); bm = fmaxf(bm, __shfl_xor(bm, 32));
                if (bm > m) {                                           // wave-uniform (every lane holds the same bm and m)
                    const float a = __builtin_amdgcn_exp2f((m - bm) * LOG2E);       // first block: exp2(-inf) = 0 on empty accumulators
                    l = l * a;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) { o[ct][0] *= a; o[ct][1] *= a; o[ct][2] *= a; o[ct][3] *= a; }
                    m = bm;
                }
                const float mc = m * LOG2E;
                f16x8_t pb;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    // masked keys: exp2(-inf) = 0; the sum is of the rounded values the MFMA multiplies
                    for (int r = 0; r < 4; ++r) { const half_t ph = f2h(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], LOG2E, -mc))); pb[4 * t + r] = ph; l = l + h2f(ph); }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ring[j][4 + ct]), pb, o[ct], 0, 0, 0);
                if (n + RD < cnt) issue(ring[j], kb + RD * step);              // the slot's next block (RD - 1 blocks stay in flight while one is consumed)
            }
        }
    }
    l = l + __shfl_xor(l, 16); l = l + __shfl_xor(l, 32);
    float* cw = cmb[hs][part];
    if (lane == 0) { cw[0] = m; cw[1] = l; }
    if (r16 == 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) cw[2 + ct * 16 + 4 * g + r] = o[ct][r];
    }
    __syncthreads();
    if (part == 0 && valid) {
        float M = cmb[hs][0][0];
#pragma unroll
        for (int i = 1; i < 4; ++i) M = fmaxf(M, cmb[hs][i][0]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        // an empty quarter: m = -inf, a = 0
        for (int i = 0; i < 4; ++i) { const float a = __builtin_amdgcn_exp2f((cmb[hs][i][0] - M) * LOG2E);
        num = __builtin_fmaf(cmb[hs][i][2 + lane], a, num); den = __builtin_fmaf(cmb[hs][i][1], a, den); }
        att_store_m(out, b, ldo, h * 64 + lane, num / den, f32_out, ofrag_k);
    }
    if (rec) {      // earliest begin and latest end over the workgroups, off the critical path (thread 0 sits in a wave that does the final combine and store)
        atomicMax(&rec->t0_inv, ~t_in); atomicMax(&rec->t1, wall_clock64());
        if (blockIdx.x == 0) atomicAdd(&rec->live_rows, 1u);
    }
}
void skw_dec_cross_attn_vt(const half_t* q, const half_t* ck, const half_t* cvt, int B, int H, int d, int n_ctx, int Tpad, half_t* out, const int* active, hipStream_t s,
    int f32_out, int pv16, const int* seq, hipEvent_t ev_start, hipEvent_t ev_stop, int ofrag, SkwKClk* clk) {
    const int as = (int)(sizeof(SkwSeqState) / 4);
    const dim3 grid((H + 2) / 3, B), blk(768);        // three heads per workgroup: 256 workgroups of 12 waves at 64 rows x 12 heads, one per CU
    // (with events: hipExtLaunchKernelGGL stamps them at the kernel's own begin and end — the duration rocprofv3 reports — instead of an event pair around the launch, which adds the dispatch gap)
    if (pv16 == 2) {          // f16_mfma precision, fragment-order cross K / V^T: the one-pass streaming kernel
        if (ev_start) hipExtLaunchKernelGGL((k_dec_cross_attn16<3, 3>), grid, blk, 0, s, ev_start, ev_stop, 0, q, (long)d, ck, (long)Tpad * d, (long)d, cvt, n_ctx, Tpad, H, out, (long)d,
            active, as, f32_out & 1, seq, ofrag ? d : 0, clk);
        else hipLaunchKernelGGL((k_dec_cross_attn16<3, 3>), grid, blk, 0, s, q, (long)d, ck, (long)Tpad * d, (long)d, cvt, n_ctx, Tpad, H, out, (long)d, active, as, f32_out & 1, seq,
            ofrag ? d : 0, clk);
        return;
    }
    // row layouts: the two-phase kernel; pv16 == 1 (f16_mfma with SKW_XATTN_FRAG=0) takes a 32-key block of P.V in one f16 MFMA, the exact precision chains f32 MFMAs key by key
    if (pv16 && ev_start) hipExtLaunchKernelGGL((k_dec_cross_attn<24, 4, 3, true>), grid, blk, 0, s, ev_start, ev_stop, 0, q, (long)d, ck, (long)n_ctx * d, (long)d, cvt, n_ctx, Tpad, H, out,
        (long)d, active, as, f32_out & 1, seq);
    else if (ev_start) hipExtLaunchKernelGGL((k_dec_cross_attn<24, 4, 3>), grid, blk, 0, s, ev_start, ev_stop, 0, q, (long)d, ck, (long)n_ctx * d, (long)d, cvt, n_ctx, Tpad, H, out,
        (long)d, active, as, f32_out & 1, seq);
    else if (pv16) hipLaunchKernelGGL((k_dec_cross_attn<24, 4, 3, true>), grid, blk, 0, s, q, (long)d, ck, (long)n_ctx * d, (long)d, cvt, n_ctx, Tpad, H, out, (long)d, active, as, f32_out & 1, seq);
    else hipLaunchKernelGGL((k_dec_cross_attn<24, 4, 3>), grid, blk, 0, s, q, (long)d, ck, (long)n_ctx * d, (long)d, cvt, n_ctx, Tpad, H, out, (long)d, active, as, f32_out & 1, seq);
}

void skw_dec_self_attn(const half_t* q, const half_t* kc, const half_t* vc, const int* pos, int B, int H, int d, int n_text_ctx, half_t* out, const int* active,
    hipStream_t s, int f32_out, SkwQ8Out q8, const int* seq, int fastv, int ofrag) {
    if (fastv && skw_sw(SW_DEC_ATTN_FASTV) && !q8.q && !f32_out) { hipLaunchKernelGGL((k_dec_attn<7, true>), dim3((H + 3) / 4, B), dim3(256), 0, s, q, (long)d, kc, vc, (long)n_text_ctx * d, (long)d,
                       pos, (int)(sizeof(SkwSeqState) / 4), 0, H, out, (long)d, active, (int)(sizeof(SkwSeqState) / 4), f32_out, q8, seq, ofrag ? d : 0); return; }
    hipLaunchKernelGGL((k_dec_attn<7>), dim3((H + 3) / 4, B), dim3(256), 0, s, q, (long)d, kc, vc, (long)n_text_ctx * d, (long)d,
                       pos, (int)(sizeof(SkwSeqState) / 4), 0, H, out, (long)d, active, (int)(sizeof(SkwSeqState) / 4), f32_out, q8, seq, (ofrag && !q8.q && !f32_out) ? d : 0);
}

// ------------------------------------------------------------------ K11: logits -> token (+ state update)
struct ArgBest { float v; int i; };
__device__ __forceinline__ ArgBest better(ArgBest a, ArgBest b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ ArgBest block_argbest(ArgBest x, ArgBest* sh) {
    for (int o = 32; o > 0; o >>= 1) { ArgBest y; y.v = __shfl_xor(x.v, o, 64); y.i = __shfl_xor(x.i, o, 64); x = better(x, y); }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = x;
    __syncthreads();
    ArgBest r = sh[0]; for (int i = 1; i < nw; ++i) r = better(r, sh[i]);
    __syncthreads();
    return r;
}
// log-softmax statistics of the admissible logits in [lo,hi): returns max and logsumexp
__device__ void block_lse(const float* lg, int lo, int hi, float* sh_f, double* sh_d, float* out_max, float* out_lse) {
    float m = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) m = fmaxf(m, lg[i]);
    m = block_max(m, sh_f);
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) { float v = lg[i]; if (v > -INFINITY) acc += (double)skw_expf(v - m); }
    acc = block_sum_f64(acc, sh_d);
    *out_max = m;
    *out_lse = (acc > 0.0) ? skw_logf((float)acc) + m : -INFINITY;
}

// std::mt19937 / std::generate_canonical<double,53> / std::discrete_distribution as libstdc++ implements them (restated in
// oracle/skw_oracle.c, pinned there against the real library); run by one lane: the f64 sums are sequential by definition.
__global__ void k_rng_seed(uint32_t* rng_all, uint32_t seed) {
    uint32_t* mt = rng_all + (long)blockIdx.x * SKW_RNG_WORDS;
    mt[0] = seed; for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    mt[624] = 624;
}
void skw_rng_seed(uint32_t* rng, int n_clips, uint32_t seed, hipStream_t s) { hipLaunchKernelGGL(k_rng_seed, dim3(n_clips), dim3(1), 0, s, rng, seed); }
__device__ uint32_t mt_next(uint32_t* mt) {
    int idx = (int)mt[624];
    if (idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = mt[idx]; mt[624] = (uint32_t)(idx + 1);
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
__device__ int discrete_draw(const float* probs, int n, uint32_t* mt) {
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += (double)probs[i];
    double cs = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) { cs += (double)mt_next(mt) * tmp; tmp *= 4294967296.0; }
    double u = cs / tmp; if (u >= 1.0) u = 0.99999999999999988897769753748;   // nextafter(1.0, 0.0)
    double cp = 0.0;
    for (int i = 0; i < n; ++i) { cp += (double)probs[i] / sum; if (i == n - 1) cp = 1.0; if (!(cp < u)) return i; }
    return n - 1;
}

// Trace / teacher-forced decisions (skw_full_batch_traced): record what this precision chose and, when `forced` names another token, feed that one —
// the state below then evolves exactly as in the run the forced tokens came from.  lg: the row's logits in memory (filtered in place by the streaming
// kernel and by sampled passes; raw otherwise — the same number for an admissible token).
__device__ __forceinline__ void smp_trace_step(SkwTokenOut& tk, const float* lg, float lse, int i, int max_tok, long row, const int* forced, SkwTraceStep* trace,
                                               int i1, int i2, float t1, float t2, const SkwLogitParams& p, float temperature) {
    if (!trace || i >= max_tok) return;
    int fid = forced ? forced[row * max_tok + i] : -1;
    if (fid < 0 || fid >= p.n_vocab) fid = tk.id;
    SkwTraceStep ts; ts.chosen_id = tk.id; ts.forced_id = fid; ts.top1_id = i1; ts.top2_id = (t2 > -INFINITY) ? i2 : -1; ts.top1 = t1; ts.top2 = t2;
    ts.forced_logit = lg[fid]; ts.lse = lse; ts.temperature = temperature; ts.pad = 0;
    trace[row * max_tok + i] = ts;
    if (fid != tk.id) { tk.id = fid; tk.plog = lg[fid] - lse; tk.p = skw_expf(tk.plog); }
}
// The same draw by the whole workgroup.  What libstdc++ defines is sequential — sum = p_0 + p_1 + ... in f64, q_i = p_i / sum, cp_i = q_0 + ... + q_i, the first
// cp_i >= u — and the two sums stay one lane's k-ascending chains (their roundings ARE the definition).  But one lane fetching 51 865 floats from memory one at
// a time and dividing each by the sum inside the chain made a sampled step 5.8 ms (a long-form batch with temperature fall-backs: 4.5 s of a 6.4 s call).  Here
// the workgroup stages the row through LDS (64 KB at a time), the divisions — independent — are done by all threads, and the second chain stops at the hit.
// Same bits as discrete_draw (oracle/skw_oracle.c).  lds: 8 192 floats = 4 096 doubles (static LDS stays under 64 KB).
// S + p == S under round-to-nearest whenever 0 <= p < ulp(S) / 2: a 64-element block whose largest element is below that bound cannot move the running sum,
// whatever the order inside it, and since the sum never decreases it stays immovable — such blocks are skipped WITHOUT changing a bit of the sequential result.
// (Peaked distributions — most real steps — leave a few dozen elements in the chain; a flat one keeps all 51 865.)
// 2^(exponent(s) - 53); 0 for s == 0 / subnormal: nothing is skipped then
__device__ __forceinline__ double half_ulp_f64(double s) { const int e = (int)((__double_as_longlong(s) >> 52) & 0x7ff); return e == 0 ? 0.0 : __longlong_as_double((long long)max(e - 53, 1) << 52); }
__device__ int block_discrete_draw(const float* probs, double* q, int n, uint32_t* mt, float* lds, double* s_sum, int* s_hit) {
    const int tid = threadIdx.x, nt = blockDim.x;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ double bmax[64];                                  // per 64-element block of the staged chunk
    double* ldsd = (double*)lds;                                 // the chunk as doubles: 4 096 per stage (the serial lane then issues one LDS read and two adds per two elements)
    double sum = 0.0;
    for (int c0 = 0; c0 < n; c0 += 4096) {
        const int m = min(4096, n - c0);
        for (int i = tid; i < m; i += nt) ldsd[i] = (double)probs[c0 + i];
        __syncthreads();
        if (tid < 64) { double mx = 0.0; for (int i = tid * 64; i < min(tid * 64 + 64, m); ++i) mx = fmax(mx, ldsd[i]); bmax[tid] = mx; }
        __syncthreads();
        if (tid == 0) {
            for (int b0 = 0; b0 < m; b0 += 64) {
                if (bmax[b0 >> 6] < half_ulp_f64(sum)) continue;
                const int e = min(b0 + 64, m); int i = b0;
                for (; i + 2 <= e; i += 2) { const f64x2 v = *(const f64x2*)(ldsd + i); sum += v[0]; sum += v[1]; }
                for (; i < e; ++i) sum += ldsd[i];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        double cs = 0.0, tmp = 1.0;
        for (int k = 0; k < 2; ++k) { cs += (double)mt_next(mt) * tmp; tmp *= 4294967296.0; }
        double u = cs / tmp; if (u >= 1.0) u = 0.99999999999999988897769753748;   // nextafter(1.0, 0.0)
        s_sum[0] = sum; s_sum[1] = u; *s_hit = -1;
    }
    __syncthreads();
    sum = s_sum[0]; const double u = s_sum[1];
    for (int i = tid; i < n; i += nt) q[i] = (double)probs[i] / sum;
    __threadfence_block();
    __syncthreads();
    double cp = 0.0;
    for (int c0 = 0; c0 < n; c0 += 4096) {
        const int m = min(4096, n - c0);
        for (int i = tid; i < m; i += nt) ldsd[i] = q[c0 + i];
        __syncthreads();
        if (tid < 64) { double mx = 0.0; for (int i = tid * 64; i < min(tid * 64 + 64, m); ++i) mx = fmax(mx, ldsd[i]); bmax[tid] = mx; }
        __syncthreads();
        if (tid == 0) {
            int hit = -1;
            for (int b0 = 0; b0 < m && hit < 0; b0 += 64) {
                const int e = min(b0 + 64, m); const bool last_block = c0 + e == n;
                if (!last_block && bmax[b0 >> 6] < half_ulp_f64(cp)) continue;      // (the block that holds the last element is walked: cp = 1 there by definition)
                if (!last_block) {      // cp never decreases: a block whose END is still below u holds no hit — one compare per block instead of one per element
                    double c2 = cp; int i = b0;
                    for (; i + 2 <= e; i += 2) { const f64x2 v = *(const f64x2*)(ldsd + i); c2 += v[0]; c2 += v[1]; }
                    for (; i < e; ++i) c2 += ldsd[i];
                    if (c2 < u) { cp = c2; continue; }
                }
                for (int i = b0; i < e; ++i) { cp += ldsd[i]; if (c0 + i == n - 1) cp = 1.0; if (!(cp < u)) { hit = c0 + i; break; } }
            }
            *s_hit = hit;
        }
        __syncthreads();
        if (*s_hit >= 0) break;      // (uniform)
        __syncthreads();
    }
    const int h = *s_hit;
    return h >= 0 ? h : n - 1;
}

__global__ __launch_bounds__(1024) void k_dec_sample_stream(float* logits_all, const uint8_t* static_mask, SkwLogitParams p, SkwSeqState* st_all, SkwTokenOut* toks_all,
                                                     int max_tok, int* n_active, float* probs_all, uint32_t* rng_all, const int* clip_idx, const int* prompt_buf,
                                                     const int* forced, SkwTraceStep* trace) {
    __shared__ float sh_f[16]; __shared__ double sh_d[16]; __shared__ ArgBest sh_a[16];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    SkwSeqState* st = &st_all[b];
    if (!st->active) return;   // uniform per block
    if (st->cur_pos < st->n_prompt - 1) {     // still feeding the prompt ([prev] + past text + sot/lang/task): next prompt token, no sampling (uniform per block)
        if (threadIdx.x == 0) { const int np = st->cur_pos + 1; st->cur_token = prompt_buf[(long)b * SKW_PROMPT_CAP + np]; st->cur_pos = np; }
        return;
    }
    float* lg = logits_all + (long)b * p.n_vocab;
    const int NV = p.n_vocab;
    SkwTokenOut* toks = toks_all + (long)b * max_tok;
    const int n_tok = st->n_tokens;
    const bool is_initial = n_tok == 0;
    float mx, lse;
    if (is_initial) {   // no_speech_prob from the unfiltered distribution
        block_lse(lg, 0, NV, sh_f, sh_d, &mx, &lse);
        if (tid == 0) st->no_speech_prob = skw_expf(lg[p.tok_nosp] - lse);
    }
    const float temperature = st->temperature;
    if (temperature > 0.0f) { __syncthreads(); for (int i = tid; i < NV; i += nt) lg[i] = lg[i] / temperature; __syncthreads(); }   // before any filter
    const int last_id = n_tok > 0 ? toks[n_tok - 1].id : -1;
    const int pen_id = n_tok > 1 ? toks[n_tok - 2].id : -1;
    const bool last_ts = n_tok > 0 && last_id >= p.tok_beg;
    const bool pen_ts = n_tok < 2 || pen_id >= p.tok_beg;
    const int has_ts = st->has_ts; const int ts_lo = has_ts ? p.tok_beg + st->seek_delta / 2 : 0;
    for (int i = tid; i < NV; i += nt) {
        bool kill = static_mask[i] != 0;
        if (is_initial && p.suppress_blank && (i == p.tok_eot || i == p.tok_space)) kill = true;
        if (p.no_timestamps && i >= p.tok_beg) kill = true;
        if (last_ts) { if (pen_ts) { if (i >= p.tok_beg) kill = true; } else { if (i < p.tok_eot) kill = true; } }
        if (is_initial && p.tid0_initial >= 0 && i >= p.tok_beg + p.tid0_initial + 1) kill = true;
        if (has_ts && i >= p.tok_beg && i < ts_lo) kill = true;
        if (kill) lg[i] = -INFINITY;
    }
    __syncthreads();
    block_lse(lg, 0, NV, sh_f, sh_d, &mx, &lse);
    // timestamp mass rule, on logprobs = logits - lse
    float ts_logprob = -INFINITY;
    {
        float m = -INFINITY;
        for (int i = p.tok_beg + tid; i < NV; i += nt) { float v = lg[i]; if (v > -INFINITY) m = fmaxf(m, v - lse); }
        m = block_max(m, sh_f);
        double acc = 0.0;
        for (int i = p.tok_beg + tid; i < NV; i += nt) { float v = lg[i]; if (v > -INFINITY) acc += (double)skw_expf((v - lse) - m); }
        acc = block_sum_f64(acc, sh_d);
        if (acc > 0.0) ts_logprob = skw_logf((float)acc) + m;
    }
    float max_text = -INFINITY;
    for (int i = tid; i < p.tok_beg; i += nt) { float v = lg[i]; if (v > -INFINITY) max_text = fmaxf(max_text, v - lse); }
    max_text = block_max(max_text, sh_f);
    const bool force_ts = ts_logprob > max_text;
    const int lo = force_ts ? p.tok_beg : 0;
    if (force_ts) { for (int i = tid; i < p.tok_beg; i += nt) lg[i] = -INFINITY; __syncthreads(); }
    // best token over probs = expf(logprob), first index wins ties; timestamp statistics
    ArgBest best = {0.0f, 0}, bts = {0.0f, 0x7fffffff};
    double sum_ts = 0.0; float top1 = -INFINITY, top2 = -INFINITY;
    const bool sampled = temperature > 0.0f;
    float* probs = probs_all + (long)b * skw_probs_row_floats(NV);
    if (sampled) { for (int i = tid; i < NV; i += nt) probs[i] = 0.0f; __syncthreads(); }   // (the loop below starts at lo: another thread owns the entry)
    for (int i = lo + tid; i < NV; i += nt) {
        float v = lg[i];
        if (v > -INFINITY) {
            float pr = skw_expf(v - lse);
            if (sampled) probs[i] = pr;
            ArgBest c = {pr, i}; if (pr > best.v || (pr == best.v && i < best.i)) best = c;
            if (i >= p.tok_beg) { sum_ts += (double)pr; if (pr > bts.v || (pr == bts.v && pr > 0.0f && i < bts.i)) { bts.v = pr; bts.i = i; } }
            if (v > top1) { top2 = top1; top1 = v; } else if (v > top2) top2 = v;
        }
    }
    best = block_argbest(best, sh_a);
    bts = block_argbest(bts, sh_a);
    sum_ts = block_sum_f64(sum_ts, sh_d);
    // margin between the two largest admissible logits
    float t1 = block_max(top1, sh_f);
    float cand = (top1 == t1) ? top2 : top1;   // exact duplicates of the max report a margin of 0 via top2 only within a thread; fine for diagnostics
    float t2 = block_max(cand, sh_f);
    if (sampled) { __threadfence_block(); __syncthreads(); }
    if (tid != 0) return;
    SkwTokenOut tk; tk.id = best.i; tk.p = best.v;
    if (sampled) { tk.id = discrete_draw(probs, NV, rng_all + (long)clip_idx[b] * SKW_RNG_WORDS); tk.p = probs[tk.id]; }
    tk.plog = lg[tk.id] - lse;
    const int i = n_tok;
    smp_trace_step(tk, lg, lse, i, max_tok, b, forced, trace, best.i, -1, t1, t2, p, temperature);
    tk.tid = (bts.v > 0.0f) ? bts.i : 0; tk.pt = (float)((double)bts.v / (sum_ts + 1e-10)); tk.ptsum = (float)sum_ts;
    if (tk.id >= p.tok_beg) { tk.tid = tk.id; tk.pt = tk.p; }
    tk.margin = (!sampled && t2 > -INFINITY) ? t1 - t2 : INFINITY;
    if (!sampled && t2 > -INFINITY && t1 - t2 < st->min_margin) st->min_margin = t1 - t2;   // argmax passes only (diagnostic)
    if (i < max_tok) toks[i] = tk;
    st->n_tokens = i + 1;
    int failed = 0, completed = 0;
    if (tk.id > p.tok_beg) {
        const int sd_new = 2 * (tk.id - p.tok_beg);
        if (st->has_ts && st->seek_delta > sd_new && st->result_len < i) failed = 1;
        else { st->seek_delta = sd_new; st->result_len = i + 1; st->has_ts = 1; }
    }
    if (!failed && (tk.id == p.tok_eot || (p.max_tokens > 0 && i >= p.max_tokens) || (st->has_ts && st->seek + st->seek_delta + SKW_DELTA_MIN >= st->seek_end))) {
        if (st->result_len == 0 && !p.no_timestamps) {
            if (st->seek + st->seek_delta + SKW_DELTA_MIN >= st->seek_end) st->result_len = i + 1; else failed = 1;
        }
        if (!failed) {
            if (p.single_segment || p.no_timestamps) { st->result_len = i + 1; st->seek_delta = 100 * 30; }
            completed = 1;
        }
    }
    if (!failed && !completed && i == p.n_max - 1 && (st->result_len == 0 || st->seek_delta < 100 * 30 / 2)) failed = 1;
    if (!failed && !completed && i + 1 >= p.n_max) completed = 1;   // loop bound reached (whisper.cpp leaves the for loop)
    st->failed = failed; st->completed = completed;
    st->cur_token = tk.id; st->cur_pos = st->n_prompt + i;
    if (failed || completed) { st->active = 0; n_active[b] = 0; }     // the row's live flag, in host-mapped memory: the host reads it after the stream drains (no copy kernel in the step)
}
// Register-resident form: the row's logits (<= 104 per thread) are loaded once, every pass of whisper_process_logits then runs on
// registers -- the streaming form above re-reads the row from L2 six times with nothing to overlap the latency (90 us per step).
// Same operations per logit; skw_expf(-inf) == 0, so suppressed logits need no branches; exponentials go two at a time through
// packed math; f64 partial sums are grouped differently, which the f64 accumulation makes immaterial (D1).
#define SMP_PT 104
#define SMP_NT 512      // 8 waves: 2 per SIMD, so the 104 resident logits fit the 256-VGPR budget
#define SMP_TX 96       // stripes [0, SMP_TX) hold text tokens only in every Whisper vocabulary (tok_eot = 50256 / 50257 >= 96 * 512)
struct SmpMain { ArgBest best; float best_logit; ArgBest bts; double sum_ts; float top1, top2; };
// element index of slot c: recomputed inside each pass from a value the optimiser cannot see through -- shared across passes, the
// ~100 indices would stay live for the whole kernel and push the logits into scratch
#define SMP_PASS_BEGIN { int tq = tid; asm volatile("" : "+v"(tq));
#define SMP_PASS_END }
#define SMP_IDX(c) (tq + SMP_NT * (c))
// TRACE: the trace / teacher-forced form (one more pass for the runner-up's index); DRAW: rows at a temperature > 0 may be present (the workgroup-wide draw and its LDS
//  staging are compiled in); the greedy step's graph holds <false, false>
template <bool TRACE, bool DRAW>
__global__ __launch_bounds__(SMP_NT) void k_dec_sample(float* logits_all, const uint8_t* static_mask, SkwLogitParams p, SkwSeqState* st_all, SkwTokenOut* toks_all,
                                                       int max_tok, int* n_active, float* probs_all, uint32_t* rng_all, const int* clip_idx, const int* prompt_buf,
                                                       const int* forced, SkwTraceStep* trace) {
    __shared__ float sh_f[2][SMP_NT / 64]; __shared__ double sh_d[SMP_NT / 64]; __shared__ SmpMain sh_m[SMP_NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    SkwSeqState* st = &st_all[b];
    if (!st->active) return;   // uniform per block
    if (st->cur_pos < st->n_prompt - 1) {     // still feeding the prompt ([prev] + past text + sot/lang/task): next prompt token, no sampling (uniform per block)
        if (threadIdx.x == 0) { const int np = st->cur_pos + 1; st->cur_token = prompt_buf[(long)b * SKW_PROMPT_CAP + np]; st->cur_pos = np; }
        return;
    }
    float* lg = logits_all + (long)b * p.n_vocab;
    const int NV = p.n_vocab;
    SkwTokenOut* toks = toks_all + (long)b * max_tok;
    const int n_tok = st->n_tokens;
    const bool is_initial = n_tok == 0;
    float v[SMP_PT]; unsigned long long killbits[2];
    { const unsigned long long* kw = (const unsigned long long*)(static_mask + ((NV + 15) & ~15)); killbits[0] = kw[tid]; killbits[1] = kw[SMP_NT + tid]; }
#pragma unroll
    for (int c = 0; c < SMP_PT; ++c) {
        if (SMP_NT * (c + 1) <= NV) v[c] = lg[tid + SMP_NT * c];                                  // whole stripe inside the row (uniform test)
        else { const int i = tid + SMP_NT * c; v[c] = (i < NV) ? lg[i] : -INFINITY; }
    }
    const int last_id = n_tok > 0 ? toks[n_tok - 1].id : -1;
    const int pen_id = n_tok > 1 ? toks[n_tok - 2].id : -1;
    const float temperature = st->temperature;
    const int has_ts = st->has_ts; const int ts_lo = has_ts ? p.tok_beg + st->seek_delta / 2 : 0;
    auto bmax2 = [&](float a, float c2, float* oa, float* oc) {     // two block-wide maxima with one exchange
        a = skw_wave_max_f32(a); c2 = skw_wave_max_f32(c2);
        if (lane == 0) { sh_f[0][w] = a; sh_f[1][w] = c2; }
        __syncthreads();
        float ra = sh_f[0][0], rc = sh_f[1][0];
        for (int k = 1; k < SMP_NT / 64; ++k) { ra = fmaxf(ra, sh_f[0][k]); rc = fmaxf(rc, sh_f[1][k]); }
        __syncthreads();
        *oa = ra; *oc = rc;
    };
    float dummy;
    if (is_initial) {   // no_speech_prob from the unfiltered distribution
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) m = fmaxf(m, v[c]);
        bmax2(m, m, &m, &dummy);
        double acc0 = 0.0;
#pragma unroll
        for (int c = 0; c < SMP_PT; c += 2) {
            f32x2 e = expf_nonpos_x2((f32x2){v[c], v[c + 1]} - (f32x2){m, m}); acc0 += (double)e[0]; acc0 += (double)e[1];
            if ((c & 7) == 6) __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from overlapping dozens of exponentials and spilling
        }
        acc0 = block_sum_f64(acc0, sh_d);
        const float lse0 = (acc0 > 0.0) ? skw_logf((float)acc0) + m : -INFINITY;
        SMP_PASS_BEGIN
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) if (SMP_IDX(c) == p.tok_nosp) st->no_speech_prob = skw_expf(v[c] - lse0);
        SMP_PASS_END
    }
    if (temperature > 0.0f) {
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) v[c] = v[c] / temperature;      // before any filter
    }
    const bool last_ts = n_tok > 0 && last_id >= p.tok_beg;
    const bool pen_ts = n_tok < 2 || pen_id >= p.tok_beg;
    // stripes [0, tx) hold text tokens only (every index below both tok_eot and tok_beg): there the index rules reduce to block-uniform
    // conditions plus the blank rule's tok_space, and no pass needs a per-lane index compare — 98 of the 104 stripes of the multilingual vocabulary
    constexpr int tx = SMP_TX;      // (compile-time: the index rules are then only compiled for the last SMP_PT - SMP_TX stripes; the host checks tok_eot, tok_beg >= SMP_TX * SMP_NT)
    const bool text_all_killed = last_ts && !pen_ts;            // "a lone timestamp must be followed by a timestamp": every token below tok_eot goes
    // (the blank rule's tok_space joins the static kill bits of the thread that owns it, so a text stripe's rule is one bit test: written with
    //  index compares, hipcc turned the rules into 104 scalar lane masks, spilled through v_writelane, ~50 instructions and four branches per logit)
    if (is_initial && p.suppress_blank) {
        const int cs = p.tok_space / SMP_NT;
        if (tid == p.tok_space % SMP_NT && cs < SMP_PT) { if (cs < 64) killbits[0] |= 1ull << cs; else killbits[1] |= 1ull << (cs - 64); }
    }
    if (text_all_killed) {      // (uniform)
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) if (c < tx) v[c] = -INFINITY;
    }
    SMP_PASS_BEGIN
#pragma unroll
    for (int c = 0; c < SMP_PT; ++c) {
        const int i = SMP_IDX(c);
        bool kill = (killbits[c >> 6] >> (c & 63)) & 1;
        if (c < tx) {      // (uniform)
            v[c] = kill ? -INFINITY : v[c];
            continue;
        }
        if (is_initial && p.suppress_blank && (i == p.tok_eot || i == p.tok_space)) kill = true;
        if (p.no_timestamps && i >= p.tok_beg) kill = true;
        if (last_ts) { if (pen_ts) { if (i >= p.tok_beg) kill = true; } else { if (i < p.tok_eot) kill = true; } }
        if (is_initial && p.tid0_initial >= 0 && i >= p.tok_beg + p.tid0_initial + 1) kill = true;
        if (has_ts && i >= p.tok_beg && i < ts_lo) kill = true;
        if (kill) v[c] = -INFINITY;
    }
    SMP_PASS_END
    // log-softmax statistics of the admissible logits
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < SMP_PT; ++c) mx = fmaxf(mx, v[c]);
    bmax2(mx, mx, &mx, &dummy);
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < SMP_PT; c += 2) {
        f32x2 e = expf_nonpos_x2((f32x2){v[c], v[c + 1]} - (f32x2){mx, mx}); acc += (double)e[0]; acc += (double)e[1];
        if ((c & 7) == 6) __builtin_amdgcn_sched_barrier(0);
    }
    acc = block_sum_f64(acc, sh_d);
    const float lse = (acc > 0.0) ? skw_logf((float)acc) + mx : -INFINITY;
    // timestamp mass rule, on logprobs = logits - lse
    float m_ts = -INFINITY, max_text = -INFINITY;
    SMP_PASS_BEGIN
#pragma unroll
    for (int c = 0; c < SMP_PT; ++c) { const float lp = v[c] - lse;
    if (c < tx) max_text = fmaxf(max_text, lp); else if (SMP_IDX(c) >= p.tok_beg) m_ts = fmaxf(m_ts, lp); else max_text = fmaxf(max_text, lp); }
    SMP_PASS_END
    bmax2(m_ts, max_text, &m_ts, &max_text);
    double acc_ts = 0.0;
    SMP_PASS_BEGIN
#pragma unroll
    for (int c = 0; c < SMP_PT; ++c) {
        if (c >= tx && SMP_NT * (c + 1) > p.tok_beg) {     // (uniform) only the last few stripes reach the timestamp range
            const float e = skw_expf((v[c] - lse) - m_ts);
            if (SMP_IDX(c) >= p.tok_beg) acc_ts += (double)e;
        }
    }
    SMP_PASS_END
    acc_ts = block_sum_f64(acc_ts, sh_d);
    const float ts_logprob = (acc_ts > 0.0) ? skw_logf((float)acc_ts) + m_ts : -INFINITY;
    const bool force_ts = ts_logprob > max_text;
    if (force_ts) {
        SMP_PASS_BEGIN
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) if (c < tx || SMP_IDX(c) < p.tok_beg) v[c] = -INFINITY;
        SMP_PASS_END
    }
    // The two largest admissible logits and the owner of the largest (order-free; also the margin diagnostic).  exp is increasing and
    // skw_expf follows it to a couple of ulps, so a logit more than 1e-4 below another has a strictly smaller probability (1e-4 is ~1700 ulps
    // of the ratio): when the runner-up is that far down, the winner of "largest probability, first index on ties" is the owner of
    // the largest logit, and the only exponentials the step still needs are the timestamp stripes' (their sum and their best).  Otherwise —
    // a near-tie, or a temperature pass that needs every probability — the whole row goes through the pass, as before.  Same bits either way.
    const bool sampled = DRAW && temperature > 0.0f;
    float* probs = probs_all + (long)b * skw_probs_row_floats(NV);
    float t1 = -INFINITY, t2 = -INFINITY; int i1 = 0;
    SMP_PASS_BEGIN
#pragma unroll
    // (selects, not branches; i1 counts stripes here)
    for (int c = 0; c < SMP_PT; ++c) { const float x = v[c]; const bool gt = x > t1; t2 = gt ? t1 : fmaxf(t2, x); i1 = gt ? c : i1; t1 = gt ? x : t1; }
    i1 = SMP_IDX(i1);
    SMP_PASS_END
    for (int o = 32; o > 0; o >>= 1) {
        const float o1 = __shfl_xor(t1, o, 64), o2 = __shfl_xor(t2, o, 64); const int oi = __shfl_xor(i1, o, 64);
        if (o1 > t1 || (o1 == t1 && oi < i1)) { t2 = fmaxf(t1, o2); t1 = o1; i1 = oi; } else t2 = fmaxf(t2, o1);
    }
    __shared__ float sh_t1[SMP_NT / 64], sh_t2[SMP_NT / 64]; __shared__ int sh_i1[SMP_NT / 64];
    if (lane == 0) { sh_t1[w] = t1; sh_t2[w] = t2; sh_i1[w] = i1; }
    __syncthreads();
    t1 = sh_t1[0]; t2 = sh_t2[0]; i1 = sh_i1[0];
    for (int k = 1; k < SMP_NT / 64; ++k) { const float o1 = sh_t1[k], o2 = sh_t2[k]; const int oi = sh_i1[k];
        if (o1 > t1 || (o1 == t1 && oi < i1)) { t2 = fmaxf(t1, o2); t1 = o1; i1 = oi; } else t2 = fmaxf(t2, o1); }
    int i2 = 0x7fffffff;
    if (TRACE) {      // owner of the runner-up: lowest index other than i1 that holds t2
        SMP_PASS_BEGIN
#pragma unroll
        for (int c = 0; c < SMP_PT; ++c) { const int i = SMP_IDX(c); if (v[c] == t2 && i != i1 && i < i2) i2 = i; }
        SMP_PASS_END
        for (int o = 32; o > 0; o >>= 1) i2 = min(i2, __shfl_xor(i2, o, 64));
        __syncthreads();
        if (lane == 0) sh_i1[w] = i2;
        __syncthreads();
        i2 = sh_i1[0]; for (int k = 1; k < SMP_NT / 64; ++k) i2 = min(i2, sh_i1[k]);
    }
    const bool fast = !sampled && (t1 - t2 > 1e-4f);           // (t2 == -inf: a single admissible token)
    const int c_lo = fast ? tx : 0;
    // best token over probs = expf(logprob), first index wins ties; timestamp statistics
    SmpMain r; r.best = {0.0f, 0}; r.best_logit = -INFINITY; r.bts = {0.0f, 0x7fffffff}; r.sum_ts = 0.0; r.top1 = t1; r.top2 = t2;
    SMP_PASS_BEGIN
#pragma unroll
    for (int c = 0; c < SMP_PT; c += 2) {
        if (c < c_lo) continue;      // (uniform)
        const f32x2 e2 = expf_nonpos_x2((f32x2){v[c], v[c + 1]} - (f32x2){lse, lse});     // 0 for suppressed logits: they can win nothing below
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = SMP_IDX(c + u); const float x = v[c + u]; const float pr = e2[u];
            if (pr > r.best.v || (pr == r.best.v && i < r.best.i)) { r.best.v = pr; r.best.i = i; r.best_logit = x; }
            if (c >= tx && i >= p.tok_beg) { r.sum_ts += (double)pr; if (pr > r.bts.v || (pr == r.bts.v && pr > 0.0f && i < r.bts.i)) { r.bts.v = pr; r.bts.i = i; } }
            if (sampled && i < NV) { probs[i] = pr; lg[i] = x; }     // the draw (one lane, below) walks the row in memory
        }
        if ((c & 7) == 6) __builtin_amdgcn_sched_barrier(0);
    }
    SMP_PASS_END
    if (fast && tid == 0) {      // the row's winner, wherever its stripe is: (probability, index) beats whatever the timestamp stripes hold
        const f32x2 e1 = expf_nonpos_x2((f32x2){t1, t1} - (f32x2){lse, lse});
        if (e1[0] > r.best.v || (e1[0] == r.best.v && i1 < r.best.i)) { r.best.v = e1[0]; r.best.i = i1; r.best_logit = t1; }
    }
    auto comb = [](SmpMain a, const SmpMain& c2) {
        if (c2.best.v > a.best.v || (c2.best.v == a.best.v && c2.best.i < a.best.i)) { a.best = c2.best; a.best_logit = c2.best_logit; }
        a.bts = better(a.bts, c2.bts); a.sum_ts += c2.sum_ts;
        return a;
    };
    for (int o = 32; o > 0; o >>= 1) {
        SmpMain y;
        y.best.v = __shfl_xor(r.best.v, o, 64); y.best.i = __shfl_xor(r.best.i, o, 64); y.best_logit = __shfl_xor(r.best_logit, o, 64);
        y.bts.v = __shfl_xor(r.bts.v, o, 64); y.bts.i = __shfl_xor(r.bts.i, o, 64); y.sum_ts = __shfl_xor(r.sum_ts, o, 64);
        y.top1 = t1; y.top2 = t2;
        r = comb(r, y);
    }
    if (lane == 0) sh_m[w] = r;
    if (sampled) __threadfence_block();
    __syncthreads();
    int drawn = -1;
    if constexpr (DRAW) if (sampled) {      // (uniform) every thread takes part in the draw
        __shared__ __attribute__((aligned(16))) float draw_lds[8192]; __shared__ double draw_s[2]; __shared__ int draw_hit;
        drawn = block_discrete_draw(probs, (double*)(probs + ((NV + 1) & ~1)), NV, rng_all + (long)clip_idx[b] * SKW_RNG_WORDS, draw_lds, draw_s, &draw_hit);
    }
    if (tid != 0) return;
    r = sh_m[0]; for (int k = 1; k < SMP_NT / 64; ++k) r = comb(r, sh_m[k]);
    const ArgBest best = r.best, bts = r.bts; const double sum_ts = r.sum_ts;
    SkwTokenOut tk; tk.id = best.i; tk.p = best.v; tk.plog = r.best_logit - lse;
    if (sampled) { tk.id = drawn; tk.p = probs[tk.id]; tk.plog = lg[tk.id] - lse; }
    const int i = n_tok;
    if (TRACE) smp_trace_step(tk, lg, lse, i, max_tok, b, forced, trace, i1, i2, t1, t2, p, temperature);
    tk.tid = (bts.v > 0.0f) ? bts.i : 0; tk.pt = (float)((double)bts.v / (sum_ts + 1e-10)); tk.ptsum = (float)sum_ts;
    if (tk.id >= p.tok_beg) { tk.tid = tk.id; tk.pt = tk.p; }
    tk.margin = (!sampled && t2 > -INFINITY) ? t1 - t2 : INFINITY;
    if (!sampled && t2 > -INFINITY && t1 - t2 < st->min_margin) st->min_margin = t1 - t2;   // argmax passes only (diagnostic)
    if (i < max_tok) toks[i] = tk;
    st->n_tokens = i + 1;
    int failed = 0, completed = 0;
    if (tk.id > p.tok_beg) {
        const int sd_new = 2 * (tk.id - p.tok_beg);
        if (st->has_ts && st->seek_delta > sd_new && st->result_len < i) failed = 1;
        else { st->seek_delta = sd_new; st->result_len = i + 1; st->has_ts = 1; }
    }
    if (!failed && (tk.id == p.tok_eot || (p.max_tokens > 0 && i >= p.max_tokens) || (st->has_ts && st->seek + st->seek_delta + SKW_DELTA_MIN >= st->seek_end))) {
        if (st->result_len == 0 && !p.no_timestamps) {
            if (st->seek + st->seek_delta + SKW_DELTA_MIN >= st->seek_end) st->result_len = i + 1; else failed = 1;
        }
        if (!failed) {
            if (p.single_segment || p.no_timestamps) { st->result_len = i + 1; st->seek_delta = 100 * 30; }
            completed = 1;
        }
    }
    if (!failed && !completed && i == p.n_max - 1 && (st->result_len == 0 || st->seek_delta < 100 * 30 / 2)) failed = 1;
    if (!failed && !completed && i + 1 >= p.n_max) completed = 1;   // loop bound reached (whisper.cpp leaves the for loop)
    st->failed = failed; st->completed = completed;
    st->cur_token = tk.id; st->cur_pos = st->n_prompt + i;
    if (failed || completed) { st->active = 0; n_active[b] = 0; }     // the row's live flag, in host-mapped memory: the host reads it after the stream drains (no copy kernel in the step)
}
size_t skw_static_mask_bytes(int n_vocab) { return (size_t)((n_vocab + 15) & ~15) + 2 * SMP_NT * sizeof(unsigned long long); }
void skw_static_mask_pack(const uint8_t* mask, int n_vocab, uint8_t* out) {
    memset(out, 0, skw_static_mask_bytes(n_vocab)); memcpy(out, mask, n_vocab);
    unsigned long long* kw = (unsigned long long*)(out + ((n_vocab + 15) & ~15));
    for (int i = 0; i < n_vocab && i < SMP_PT * SMP_NT; ++i) if (mask[i]) { const int t = i % SMP_NT, c = i / SMP_NT; kw[(c >> 6) * SMP_NT + t] |= 1ull << (c & 63); }
}
static int g_force_stream_sampler = 0;      // tests: run the streaming form where the register-resident one would (it leaves the filtered row in memory)
void skw_debug_force_stream_sampler(int on) { g_force_stream_sampler = on; }
void skw_dec_sample(float* logits, const uint8_t* static_mask, SkwLogitParams p, SkwSeqState* st, SkwTokenOut* toks, int max_tok, int B, int* n_active,
                    float* probs, uint32_t* rng, const int* clip_idx, const int* prompt_buf, hipStream_t s, const int* forced, SkwTraceStep* trace) {
    if (!g_force_stream_sampler && p.n_vocab <= SMP_PT * SMP_NT && std::min(p.tok_eot, p.tok_beg) >= SMP_TX * SMP_NT) {
        if (trace) hipLaunchKernelGGL((k_dec_sample<true, true>), dim3(B), dim3(SMP_NT), 0, s, logits, static_mask, p, st, toks, max_tok, n_active, probs, rng, clip_idx, prompt_buf, forced, trace);
        else if (p.any_sampled) hipLaunchKernelGGL((k_dec_sample<false, true>), dim3(B), dim3(SMP_NT), 0, s, logits, static_mask, p, st, toks, max_tok, n_active, probs,
            rng, clip_idx, prompt_buf, nullptr, nullptr);
        else hipLaunchKernelGGL((k_dec_sample<false, false>), dim3(B), dim3(SMP_NT), 0, s, logits, static_mask, p, st, toks, max_tok, n_active, probs, rng, clip_idx, prompt_buf, nullptr, nullptr);
    } else hipLaunchKernelGGL(k_dec_sample_stream, dim3(B), dim3(1024), 0, s, logits, static_mask, p, st, toks, max_tok, n_active, probs, rng, clip_idx, prompt_buf, forced, trace);
}

// ------------------------------------------------------------------ R1: audio::resampler arithmetic (rubato FastFixedIn, Linear)
// Restates /root/reference/crates/nodes/src/audio/filters/resampler.rs:231-244, 384-514 (rubato 0.16.2 asynchro_fast.rs):
// per chunk: idx starts at last_index, `while idx < end_idx { idx += t_ratio; out = (1-frac)*y[floor idx] + frac*y[floor idx + 1] }`,
// last_index = idx - chunk.  The index recurrence is a sequential f64 accumulation whose roundings decide `frac`, across the whole
// stream.  It is parallelised without giving that up:
//   k_resample_starts* one lane proposes every chunk's start index, output count and output offset: first by a closed form (one
//                      division per chunk; right whenever every addition of the walk is exact), and if that fails its check by
//                      stepping whole binades of the f64 index at a time (exact integer arithmetic per binade, ties included);
//   k_resample_walk    one lane PER CHUNK walks its chunk with the real f64 recurrence from the proposed start, writes
//                      (sample index, frac) for its outputs and checks that it produced the proposed count and hands the next chunk
//                      exactly the proposed start — by induction the proposal then IS the sequential walk, bit for bit;
//   k_resample_scan    the original single-lane walk, which only runs when a check failed or the proposal met a rounding tie.
// The interpolation (k_resample_lerp) is data parallel.
// The proposal must reproduce n dependent f64 additions exactly.  Inside one binade [2^e, 2^(e+1)) every x is a multiple of
// ulp_e = 2^(e-52), so fl(x + t) = x + RN_e(t), t rounded to that grid: a whole binade is one integer multiply-add.  When t sits
// exactly halfway between two grid points (44.1 kHz sources: t = 441/160 has one bit below the grid of [4, 8)) round-to-even picks
// the step that makes the result even: from an odd x that is one real addition, after which x is even and the step is the even
// neighbour for the rest of the binade.  A chunk crosses ~10 binades (index -10 .. 950), each crossing is one real f64 addition;
// 1500 chunks cost one lane well under a millisecond instead of ~30 ms.  (tests/test_cpu_frontend.py checks the same stepping against the sequential walk for 11 source
// rates x 3000 chunks on the CPU; on the device k_resample_walk checks every launch.)
__device__ __forceinline__ long long ceil_div_pos(long long a, long long b) {      // a > 0, b > 0, a < 2^62: no 64-bit integer division (software, ~1 us for one lane)
    long long q = (long long)((double)a / (double)b);
    while (q * b < a) ++q;
    while ((q - 1) * b >= a) --q;
    return q;
}
// first proposal: the closed form n = ceil((end - s) / t), s' = (s + n t) - chunk — one division per chunk, exact whenever every
// addition of the walk is exact (48 / 32 / 96 kHz -> 16 kHz: t is a small integer), which k_resample_walk then proves
__global__ void k_resample_starts_simple(double last_index, double t_ratio, int chunk, int n_chunks, double* start, int* count, int* offset, int* flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double end_idx = (double)(chunk - 9) - ceil(t_ratio);
    double s = last_index; int off = 0;
    for (int c = 0; c < n_chunks; ++c) {
        int n = 0;
        if (s < end_idx) { n = (int)ceil((end_idx - s) / t_ratio); if (n < 1) n = 1; while (n > 1 && s + (double)(n - 1) * t_ratio >= end_idx) --n; while (s + (double)n * t_ratio < end_idx) ++n; }
        start[c] = s; count[c] = n; offset[c] = off; off += n;
        s = (s + (double)n * t_ratio) - (double)chunk;
    }
    start[n_chunks] = s; offset[n_chunks] = off; *flag = 0;
}
// second proposal (runs only when the first one failed its check: *flag != 0)
__global__ void k_resample_starts(double last_index, double t_ratio, int chunk, int n_chunks, double* start, int* count, int* offset, int* flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!*flag) return;
    const double end_idx = (double)(chunk - 9) - ceil(t_ratio);
    double s = last_index; int off = 0;
    for (int c = 0; c < n_chunks; ++c) {
        double x = s; int n = 0;
        while (x < end_idx && x < 4.0) { x += t_ratio; n++; }               // negative / small indices: the few plain additions the walk makes too
        while (x < end_idx) {
            const long long bits = __double_as_longlong(x);
            const int e = (int)((bits >> 52) & 0x7ff) - 1023;                 // x in [2^e, 2^(e+1)), e >= 2
            long long xi = (bits & 0xfffffffffffffLL) | (1LL << 52);          // x / ulp_e
            const double ts = ldexp(t_ratio, 52 - e);                         // t / ulp_e, exact
            const double fl = floor(ts), fr = ts - fl;
            long long ti = (long long)fl;
            if (fr == 0.5) {
                if (xi & 1) { x = x + t_ratio; n++; continue; }               // odd x on a tie: the hardware rounds this one
                ti += ti & 1;                                                 // the even neighbour
            } else if (fr > 0.5) ti += 1;
            if (ti < 1) { x = x + t_ratio; n++; continue; }                   // (degenerate ratios: plain additions)
            const long long Bi = 1LL << 53, endi = (long long)ldexp(end_idx, 52 - e);
            const long long in_binade = ceil_div_pos(Bi - xi, ti) - 1;        // additions whose result stays below 2^(e+1)
            const long long to_end = xi < endi ? ceil_div_pos(endi - xi, ti) : 0;   // additions the loop condition still allows
            const long long k = in_binade < to_end ? in_binade : to_end;
            xi += k * ti; n += (int)k; x = ldexp((double)xi, e - 52);
            if (k == to_end) break;
            x = x + t_ratio; n++;                                             // the addition that crosses into the next binade: rounded by the hardware
        }
        start[c] = s; count[c] = n; offset[c] = off; off += n;
        s = x - (double)chunk;
    }
    start[n_chunks] = s; offset[n_chunks] = off; *flag = 2;        // 2: this proposal is the one to check
}
__global__ void k_resample_walk(const double* start, const int* count, const int* offset, double t_ratio, int chunk, int n_chunks, int* pos, float* frac,
                                int* n_out, double* last_index_out, int cap, int* flag, int* fail, int level) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    if (level == 2 && *flag != 2) return;                                      // the first proposal was proven: nothing to redo
    const double end_idx = (double)(chunk - 9) - ceil(t_ratio);
    double idx = start[c]; int n = 0; const int o = offset[c];
    while (idx < end_idx) {
        idx += t_ratio;
        const double fl = floor(idx);
        if (o + n < cap) { pos[o + n] = c * chunk + (int)fl; frac[o + n] = (float)(idx - fl); }
        n++;
    }
    idx = idx - (double)chunk;
    if (n != count[c] || idx != start[c + 1]) atomicOr(fail, 1);      // the proposal is not the sequential walk: the next level redoes it
    if (c == n_chunks - 1) { *n_out = o + n; *last_index_out = idx; }
}
__global__ void k_resample_scan(double last_index, double t_ratio, int chunk, int n_chunks, int* pos, float* frac, int* n_out, double* last_index_out, int cap, const int* flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (flag && !*flag) return;
    const double end_idx = (double)(chunk - 9) - ceil(t_ratio);
    double idx = last_index; int n = 0;
    for (int c = 0; c < n_chunks; ++c) {
        while (idx < end_idx) {
            idx += t_ratio;
            const double fl = floor(idx);
            if (n < cap) { pos[n] = c * chunk + (int)fl; frac[n] = (float)(idx - fl); }
            n++;
        }
        idx = idx - (double)chunk;
    }
    *n_out = n; *last_index_out = idx;
}
// in: [16*ch history | n_chunks*chunk*ch] interleaved (history = the 16 frames preceding the first chunk); out interleaved
__global__ void k_resample_lerp(const float* in, int channels, const int* pos, const float* frac, const int* n_out, float* out, int cap) {
    const int n = min(*n_out, cap);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * channels; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i / channels), c = (int)(i % channels);
        const float fr = frac[f]; const float* y = in + ((long)(pos[f] + 16)) * channels + c;
        out[i] = (1.0f - fr) * y[0] + fr * y[channels];
    }
}
void skw_resample_linear_launch(const float* in, int channels, double last_index, double t_ratio, int chunk, int n_chunks, int* pos, float* frac, int* n_out, double* last_index_out,
                                float* out, int cap, double* start, int* count, int* offset, int* flag, hipStream_t s, bool host_proposal) {
    const bool force_scan = skw_sw(SW_RESAMPLE_SCAN) != 0;     // the single-lane walk only (the fallback of both proposals; tests hold it to the same bits)
    if (force_scan) {
        // both flags set: "both proposals failed" is what this path stands for (skw_dsp_last_scan_fallback reports 2).  They were left as the allocation held them before round 5's
        // last day: the switch test passed or failed with whatever hipMalloc returned.
        hipMemsetAsync(flag, 0xFF, 2 * sizeof(int), s);
        hipLaunchKernelGGL(k_resample_scan, dim3(1), dim3(64), 0, s, last_index, t_ratio, chunk, n_chunks, pos, frac, n_out, last_index_out, cap, (const int*)nullptr);
    }
    else {
        // flag[0]: 0 = first proposal stands, 1 = it failed, 2 = second proposal to be checked; flag[1]: the second one failed too
        int* fail2 = flag + 1;
        hipMemsetAsync(flag, 0, 2 * sizeof(int), s);
        // first proposal: the closed form on the device, or — an inexact step over many chunks — the chunk starts the host walked and uploaded (skw_resample_linear)
        if (!host_proposal) hipLaunchKernelGGL(k_resample_starts_simple, dim3(1), dim3(64), 0, s, last_index, t_ratio, chunk, n_chunks, start, count, offset, flag);
        hipLaunchKernelGGL(k_resample_walk, dim3((n_chunks + 63) / 64), dim3(64), 0, s, start, count, offset, t_ratio, chunk, n_chunks, pos, frac, n_out, last_index_out, cap, flag, flag, 1);
        hipLaunchKernelGGL(k_resample_starts, dim3(1), dim3(64), 0, s, last_index, t_ratio, chunk, n_chunks, start, count, offset, flag);
        hipLaunchKernelGGL(k_resample_walk, dim3((n_chunks + 63) / 64), dim3(64), 0, s, start, count, offset, t_ratio, chunk, n_chunks, pos, frac, n_out, last_index_out, cap, flag, fail2, 2);
        hipLaunchKernelGGL(k_resample_scan, dim3(1), dim3(64), 0, s, last_index, t_ratio, chunk, n_chunks, pos, frac, n_out, last_index_out, cap, (const int*)fail2);
    }
    hipLaunchKernelGGL(k_resample_lerp, dim3(256), dim3(256), 0, s, in, channels, pos, frac, n_out, out, cap);
}

// ------------------------------------------------------------------ polyphase windowed-sinc resampler (quality mode; north_star's "polyphase resampler")
// out[n] = sum_t h[phase(n)][t] * x[base(n) + t - (T/2 - 1)], rational ratio L/M, T taps per phase, coefficients [L][T] from the host.
// One workgroup = PP_TILE consecutive output frames (all channels): the input span they read and, when it fits, the whole coefficient
// table are staged in LDS (coalesced 4-byte loads; table rows padded to an odd stride so that lanes on different phases hit different
// banks); per output the phase and base come from 32-bit arithmetic on the tile's (base0, phase0) — the one 64-bit division is per
// workgroup — and the tap loop is fma on two LDS operands.  Inputs are addressed absolutely: `in` holds frames [in_base, in_base + n_in)
// of the stream and everything outside [0, n_total) reads as zero, so a stream can be filtered packet by packet from a device-resident
// tail (skw_polyphase_stream_*) with results identical to filtering it whole.
#define PP_TILE 256
__global__ __launch_bounds__(256) void k_resample_polyphase(const float* in, long in_base, long n_in, long n_total, int channels, const float* coef, int L, int M, int T, int coef_in_lds,
                                                            float* out, long out_first, long n_out) {
    extern __shared__ float pp_lds[];
    const int tid = threadIdx.x, Ts = T | 1;
    const long o0 = out_first + (long)blockIdx.x * PP_TILE;
    const long num0 = o0 * (long)M; const long base0 = num0 / L; const int ph0 = (int)(num0 % L);
    const int span = (int)(((long)ph0 + (long)(PP_TILE - 1) * M) / L) + T;                 // frames [base0 - (T/2 - 1), base0 - (T/2 - 1) + span)
    const long first = base0 - (T / 2 - 1);
    float* xs = pp_lds; float* hs = pp_lds + (size_t)span * channels;
    for (int i = tid; i < span * channels; i += 256) {
        const long f = first + i / channels; const int c = i % channels;
        xs[i] = (f >= 0 && f < n_total && f >= in_base && f < in_base + n_in) ? in[(f - in_base) * channels + c] : 0.0f;
    }
    if (coef_in_lds) for (int i = tid; i < L * T; i += 256) hs[(i / T) * Ts + i % T] = coef[i];
    __syncthreads();
    const long o = o0 + tid;
    if (tid >= PP_TILE || o >= out_first + n_out) return;
    const int num = ph0 + tid * M; const int rel = num / L, ph = num % L;                  // 32-bit: ph0 < L <= 4096, tid < 256, M <= 4096
    const float* h = coef_in_lds ? hs + ph * Ts : coef + (long)ph * T;
    for (int c = 0; c < channels; ++c) {
        float acc = 0.0f; const float* x = xs + (size_t)rel * channels + c;
#pragma unroll 8
        for (int t = 0; t < T; ++t) acc = __builtin_fmaf(h[t], x[(size_t)t * channels], acc);
        out[(o - out_first) * channels + c] = acc;
    }
}
// outputs [out_first, out_first + n_out) of the stream into out[0 ..]; in = frames [in_base, in_base + n_in), n_total = frames of the stream known so far (beyond: zeros)
void skw_resample_polyphase_launch(const float* in, long in_base, long n_in, long n_total, int channels, const float* coef, int L, int M, int T, float* out,
    long out_first, long n_out, hipStream_t s) {
    if (n_out <= 0) return;
    const int span_max = (int)(((long)(L - 1) + (long)(PP_TILE - 1) * M) / L) + T;
    const size_t x_bytes = (size_t)span_max * channels * 4, h_bytes = (size_t)L * (T | 1) * 4;
    const int coef_in_lds = (x_bytes + h_bytes <= 96 * 1024) ? 1 : 0;
    if (x_bytes + (coef_in_lds ? h_bytes : 0) > 64 * 1024) {      // more dynamic LDS than the default limit: raise it on this device (per device: several GPUs may run in one process)
        static std::atomic<bool> raised[64]; int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!raised[dev].load(std::memory_order_acquire)) { hipFuncSetAttribute((const void*)k_resample_polyphase, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised[dev].store(true, std::memory_order_release); }
    }
    hipLaunchKernelGGL(k_resample_polyphase, dim3((unsigned)((n_out + PP_TILE - 1) / PP_TILE)), dim3(256), x_bytes + (coef_in_lds ? h_bytes : 0), s,
                       in, in_base, n_in, n_total, channels, coef, L, M, T, coef_in_lds, out, out_first, n_out);
}
