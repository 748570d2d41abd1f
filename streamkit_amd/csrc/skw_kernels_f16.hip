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

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA16X32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ------------------------------------------------------------------ big-M GEMM, f16 MFMA
// C[m][n] = sum_k A[m][k] * W[n][k].  Block tile BM x BN, K step 64 (128-byte LDS rows), NWM x NWN waves, each owning a
// (BM/NWM) x (BN/NWN) sub-tile of 16x16 MFMA tiles.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, one
// 1-KiB piece = 8 rows x 128 B per wave-instruction); the LDS image is lane-linear, so the bank swizzle (16-byte chunk c of row
// r sits at chunk position c ^ (r & 7)) is applied to the per-lane SOURCE address and again on the fragment reads.  Two LDS
// buffers, one barrier per K step: the DMA of step k+1 flies under the MFMAs of step k.
template <int EPI, int BM, int BN, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64, (NWM * NWN) / 4) void k_gemm16(SkwGemmArgs a) {
    constexpr int NW = NWM * NWN, WTM = BM / NWM, WTN = BN / NWN, TM = WTM / 16, TN = WTN / 16;
    constexpr int A_PIECES = BM / 8 / NW, B_PIECES = BN / 8 / NW;      // 1-KiB pieces per wave per K step
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "whole pieces per wave");
    __shared__ __attribute__((aligned(1024))) char lds[2 * (BM + BN) * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (a.N + BN - 1) / BN, nbm = (a.M + BM - 1) / BM, nblk = nbn * nbm;
    int bid = blockIdx.x;
    { int q = nblk >> 3, r = nblk & 7, x = bid & 7, y = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y; }   // XCD-aware, bijective
    const int bm = bid / nbn, bn = bid % nbn;       // the n-tiles of one m-tile run back to back on one XCD: the A panel is read from HBM once
    const int m0 = bm * BM, n0 = bn * BN;
    const int wr = wave / NWN, wc = wave % NWN, r16 = lane & 15, g = lane >> 4;

    // staging addresses: piece q covers tile rows 8q .. 8q+7; lane -> row 8q + (lane >> 3), chunk position lane & 7
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const half_t* gA[A_PIECES]; const half_t* gB[B_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        int gm = m0 + (wave * A_PIECES + i) * 8 + prow; if (gm > a.M - 1) gm = a.M - 1;     // rows past M are computed on a copy of the last row and never stored
        const long off = a.a_rows_per_batch ? (long)(gm / a.a_rows_per_batch) * a.a_batch_stride + (long)(gm % a.a_rows_per_batch) * a.lda : (long)gm * a.lda;
        gA[i] = a.A + off + pchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        int gn = n0 + (wave * B_PIECES + i) * 8 + prow; if (gn > a.N - 1) gn = a.N - 1;
        gB[i] = a.W + (long)gn * a.ldw + pchunk * 8;
    }
    auto stage = [&](int buf, int kb) {
        char* base = lds + buf * (BM + BN) * 128;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(gA[i] + kb * 64), (lptr_t)(base + (wave * A_PIECES + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(gB[i] + kb * 64), (lptr_t)(base + BM * 128 + (wave * B_PIECES + i) * 1024), 16, 0, 0);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment read offsets inside a buffer: row * 128 + ((chunk ^ (row & 7)) << 4), chunk = 4 * khalf + g; row & 7 == r16 & 7
    const int fo0 = ((g ^ (r16 & 7)) << 4), fo1 = fo0 ^ 64;
    const int aoff = (wr * WTM + r16) * 128, boff = BM * 128 + (wc * WTN + r16) * 128;
    const int nk = a.K >> 6;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
        const char* base = lds + (kb & 1) * (BM + BN) * 128;
        if (kb + 1 < nk) stage((kb + 1) & 1, kb + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? fo1 : fo0;
            f16x8 fa[TM], fb[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) fb[t] = *(const f16x8*)(base + boff + t * 2048 + fo);
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = *(const f16x8*)(base + aoff + t * 2048 + fo);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MFMA16X32(fa[i], fb[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * WTM + i * 16 + g * 4 + r, n = n0 + wc * WTN + j * 16 + r16;
                if (m < a.M && n < a.N) epi_store<EPI>(a, m, n, acc[i][j][r]);
            }
}

template <int EPI> static void launch_gemm16(const SkwGemmArgs& a, hipStream_t s) {
    // tile choice: 256 x 256 (8 waves) when both extents fill it, 128 x 128 (4 waves, two blocks per CU) otherwise
    static const int force = getenv("SKW_GEMM16_TILE") ? atoi(getenv("SKW_GEMM16_TILE")) : 0;
    const bool big = force ? force == 256 : (a.M >= 256 && a.N >= 256 && a.M % 256 == 0 && a.N % 256 == 0);
    if (big) hipLaunchKernelGGL((k_gemm16<EPI, 256, 256, 2, 4>), dim3(((a.M + 255) / 256) * ((a.N + 255) / 256)), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k_gemm16<EPI, 128, 128, 2, 2>), dim3(((a.M + 127) / 128) * ((a.N + 127) / 128)), dim3(256), 0, s, a);
}
// f16-MFMA form of skw_gemm (K must be a multiple of 64: every Whisper geometry's state, 4 x state and conv im2col widths are)
void skw_gemm16(const SkwGemmArgs& a, hipStream_t s) {
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
// per 128 queries instead of once per 16 as in the exact kernel.  Two passes over the keys: (1) S^T = K.Q^T for the exact row
// maximum; (2) S^T again, p = exp2((s - max) * scale * log2 e), row sums, and O^T += V^T . P^T with P^T taken straight from the
// S^T accumulators (MFMA row rho of a 16-key tile holds key 4 * (rho & 3) + (rho >> 2), which is the order the kperm'ed V^T rows
// store their keys in).  Normalisation by the row sum happens once, on O.
#define A16_QB 128
__global__ __launch_bounds__(256, 2) void k_attn_encoder16(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out,
                                                           int H, int n_ctx, int Tpad, float kq_scale, int qblocks) {
    __shared__ __attribute__((aligned(1024))) char lds[2][2][64 * 128];   // [buffer][K | V^T][64 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x; int bid = blockIdx.x;
    { int q = nblk >> 3, r = nblk & 7, x = bid & 7, y = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y; }   // the query blocks of one (batch, head) share an XCD's L2
    const long bh = bid / qblocks; const int qb = bid % qblocks;
    const int b = (int)(bh / H), h = (int)(bh % H);
    const int q0 = qb * A16_QB + wave * 32;
    const int r16 = lane & 15, g = lane >> 4;
    f16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qi = q0 + qt * 16 + r16; if (qi > n_ctx - 1) qi = n_ctx - 1;
        const half_t* qp = Qh + (bh * Tpad + qi) * 64 + g * 8;
        qf[qt][0] = *(const f16x8*)qp; qf[qt][1] = *(const f16x8*)(qp + 32);
    }
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(Kh + bh * Tpad * 64), 0, (unsigned)(Tpad * 64 * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(Vt + bh * 64 * Tpad), 0, (unsigned)(64 * Tpad * 2), 0x00020000);
    // staging: chunk id c = tid + 256 i -> row c >> 3, 16-byte chunk c & 7
    unsigned st_lds[2], st_k[2], st_v[2]; int st_vkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, row = c >> 3, pos = c & 7;
        st_lds[i] = (unsigned)(row * 128 + ((pos ^ (row & 7)) << 4));
        st_k[i] = (unsigned)((row * 64 + pos * 8) * 2);              // + kb * 64 rows; rows past Tpad fall outside the descriptor: zeros
        st_v[i] = (unsigned)((row * Tpad + pos * 8) * 2);            // + kb * 64 keys; chunks past Tpad are replaced by zeros below
        st_vkey[i] = pos * 8;
    }
    const int nkb = (Tpad + 63) >> 6;
    const int kappa = 4 * (r16 & 3) + (r16 >> 2);
    const int k_off = kappa * 128, k_sw = kappa & 7, v_off = r16 * 128, v_sw = r16 & 7;
    u32x4 sk[2], sv[2];
    auto load_k = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sk[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, st_k[i] + (unsigned)kb * 8192u, 0, 0);
    };
    auto load_v = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (kb * 64 + st_vkey[i] < Tpad) ? st_v[i] + (unsigned)kb * 128u : 0x7fffff00u, 0, 0);
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(&lds[buf][0][st_lds[i]]) = sk[i];
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
    // ---- pass 1: exact row maxima
    float mx0 = -INFINITY, mx1 = -INFINITY;
    load_k(0); store_k(0); __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const char* kbase = &lds[kb & 1][0][0];
        if (kb + 1 < nkb) load_k(kb + 1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s0, s1; scores(kbase, kt, s0, s1);
            if (kb == nkb - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (kb * 64 + kt * 16 + 4 * r + g >= n_ctx) { s0[r] = -INFINITY; s1[r] = -INFINITY; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { mx0 = fmaxf(mx0, s0[r]); mx1 = fmaxf(mx1, s1[r]); }
        }
        if (kb + 1 < nkb) store_k((kb + 1) & 1);
        __syncthreads();
    }
    mx0 = fmaxf(mx0, __shfl_xor(mx0, 16, 64)); mx0 = fmaxf(mx0, __shfl_xor(mx0, 32, 64));
    mx1 = fmaxf(mx1, __shfl_xor(mx1, 16, 64)); mx1 = fmaxf(mx1, __shfl_xor(mx1, 32, 64));
    // ---- pass 2: probabilities, row sums, O^T
    const float c1 = kq_scale * 1.44269504088896341f;
    const float mc0 = mx0 * c1, mc1 = mx1 * c1;
    float l0 = 0.0f, l1 = 0.0f;
    f32x4 oacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) oacc[qt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_k(0); load_v(0); store_k(0); store_v(0); __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const char* kbase = &lds[kb & 1][0][0]; const char* vbase = &lds[kb & 1][1][0];
        if (kb + 1 < nkb) { load_k(kb + 1); load_v(kb + 1); }
        f16x8 p0[2], p1[2];      // P^T fragments [32-key half of the block]: element 4 * (kt & 1) + r of lane (query, g) = key 16 kt + 4 r + g
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s0, s1; scores(kbase, kt, s0, s1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c1, -mc0)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c1, -mc1));
                if (kb == nkb - 1 && kb * 64 + kt * 16 + 4 * r + g >= n_ctx) { e0 = 0.0f; e1 = 0.0f; }
                const half_t h0 = (half_t)e0, h1 = (half_t)e1;
                l0 += (float)h0; l1 += (float)h1;                      // the sum of what P.V will actually use
                p0[kt >> 1][(kt & 1) * 4 + r] = h0; p1[kt >> 1][(kt & 1) * 4 + r] = h1;
            }
        }
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
    l0 += __shfl_xor(l0, 16, 64); l0 += __shfl_xor(l0, 32, 64);
    l1 += __shfl_xor(l1, 16, 64); l1 += __shfl_xor(l1, 32, 64);
    const float inv0 = 1.0f / l0, inv1 = 1.0f / l1;
    // O^T tiles: lane (query = r16, g) holds channels ct * 16 + 4 g + r
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + r16;
        if (qi < n_ctx) {
            half_t* op = out + ((long)b * n_ctx + qi) * ld_out;
            const float inv = qt ? inv1 : inv0;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) op[skw_kperm(h * 64 + ct * 16 + 4 * g + r)] = (half_t)(oacc[qt][ct][r] * inv);
        }
    }
}
void skw_attn_encoder16(const half_t* Qh, const half_t* Kh, const half_t* Vt, half_t* out, long ld_out, int B, int H, int n_ctx, int Tpad, hipStream_t s) {
    const int qblocks = (n_ctx + A16_QB - 1) / A16_QB;
    hipLaunchKernelGGL(k_attn_encoder16, dim3(qblocks * H * B), dim3(256), 0, s, Qh, Kh, Vt, out, ld_out, H, n_ctx, Tpad, 1.0f / sqrtf(64.0f), qblocks);
}
