// skw_kernels_q8.hip — ggml's arithmetic for block-quantised model files (q4_0 / q4_1 / q5_0 / q5_1 / q8_0), exact precision.
//
// whisper.cpp multiplies a quantised weight matrix by f32 activations the way ggml_compute_forward_mul_mat does it: every activation
// row is quantised to q8_0 / q8_1 blocks of 32 (vec_dot_type of the weight type), and each output is a block-ascending sum of
// integer block dots times f32 scales (include/skw_ggml_quant.h (a) states the operations; oracle/skw_oracle.c linear_q8 is the CPU
// restatement these kernels are bit-identical to).  The reference's DEFAULT model is a q5_1 file
// (/root/reference/plugins/native/whisper/src/lib.rs:114-116), so this is the arithmetic its default configuration runs.
//
//   k_q8_quantize   f32 rows -> int8 values [M][K] + per-block d (f16-rounded) and s = f16(d_unrounded * sum q), stored [K/32][M]
//   k_gemm_q8_lds   the encoder's products: 128 x 128 tile per workgroup, operands and scales staged through LDS by LDS-DMA (two stage buffers),
//                   per 32-block one v_mfma_i32_16x16x32_i8 per 16 x 16 sub-tile gives the sixteen-by-sixteen integer dots and the f32 update
//                   of skw_ggml_block_dot runs on the VALU, packed two features per instruction (the path is VALU-bound: ~14 instructions per
//                   integer dot of 16 x 16 x 32)
//   k_gemm_q8<TW>   the same fed from global memory (tails, geometries the staged form does not take)
//   k_gemm_q8_small the decoder's products: four waves = the four runs of the segmented block sum (D3')
// The weights are the MFMA's first operand as everywhere else: a lane ends with four adjacent features of one row.
#include <hip/hip_runtime.h>
#include "skw_dev_common.h"
#include "../../include/skw_ggml_quant.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

// one thread per block of 32 (operation for operation skw_ggml_quantize_q8_block)
__global__ __launch_bounds__(256) void k_q8_quantize(const float* x, long ldx, int M, int K, int8_t* q, float* dT, float* sT) {
    const int nb = K >> 5;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)M * nb) return;
    const int m = (int)(id / nb), b = (int)(id % nb);
    const f32x4* xp = (const f32x4*)(x + (long)m * ldx + b * 32);
    float v[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const f32x4 t = xp[j]; v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3]; }
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) { const float a = v[j] < 0.0f ? -v[j] : v[j]; if (a > amax) amax = a; }
    const float d = amax / 127.0f, idv = d != 0.0f ? 1.0f / d : 0.0f;
    int sum = 0; union { int8_t b[32]; u32x4 w[2]; } o;
#pragma unroll
    for (int j = 0; j < 32; ++j) { const float t = v[j] * idv; const int qi = (int)skw_roundf(t); o.b[j] = (int8_t)qi; sum += qi; }
    u32x4* qp = (u32x4*)(q + (long)m * K + b * 32); qp[0] = o.w[0]; qp[1] = o.w[1];
    dT[(long)b * M + m] = skw_round_f16(d);
    sT[(long)b * M + m] = skw_round_f16((float)sum * d);
}
void skw_q8_quantize(const float* x, long ldx, int M, int K, int8_t* q, float* dT, float* sT, hipStream_t s) {
    const long n = (long)M * (K >> 5);
    hipLaunchKernelGGL(k_q8_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, M, K, q, dT, sT);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// skw_ggml_block_dot for a lane's four adjacent features at once, two per packed-f32 instruction (v_pk_mul_f32 / v_pk_add_f32 round each
// element exactly as the scalar forms do; nothing is contracted): the kernels below are VALU-bound on this update
template <int FORM>
__device__ __forceinline__ void q8_update4(float (&sumf)[4], const i32x4 si, const f32x4 dw, const f32x4 mw, float dy, float sy) {
    const f32x2 s01 = {(float)si[0], (float)si[1]}, s23 = {(float)si[2], (float)si[3]};
    const f32x2 d01 = {dw[0], dw[1]}, d23 = {dw[2], dw[3]}, y2 = {dy, dy};
    f32x2 t01, t23;
    if (FORM == 1) { t01 = (s01 * d01) * y2; t23 = (s23 * d23) * y2; }
    else {
        t01 = (d01 * y2) * s01; t23 = (d23 * y2) * s23;
        if (FORM == 3) { const f32x2 m01 = {mw[0], mw[1]}, m23 = {mw[2], mw[3]}, z2 = {sy, sy}; t01 = t01 + m01 * z2; t23 = t23 + m23 * z2; }
    }
    const f32x2 a01 = (f32x2){sumf[0], sumf[1]} + t01, a23 = (f32x2){sumf[2], sumf[3]} + t23;
    sumf[0] = a01[0]; sumf[1] = a01[1]; sumf[2] = a23[0]; sumf[3] = a23[1];
}

template <int EPI>
__device__ __forceinline__ void q8_store(const SkwGemmArgs& a, int m, int n, float v) {
    if (EPI == EPI_VT_F16) epi_store<EPI_VT_F16>(a, n, m, v);          // (that epilogue names the feature first: it was written for the operand-swapped call)
    else epi_store<EPI>(a, m, n, v);
}

// C = epilogue(A_q8 . W_q^T): A int8 [M][K] with dyT / syT [K/32][M]; W int8 [N][K] with dwT / mwT [K/32][n_pad] (n_pad = N rounded up to 64).
// Workgroup tile (32 TW) x (32 TW): four waves of (16 TW) x (16 TW), each TW x TW MFMA tiles.  TW = 4 (128 x 128) is the encoder's shape: a
// wave then makes 24 loads per 32-block for 16 integer dots of 16 x 16 (TW = 2: 12 loads for 4), which is what the kernel is bound by — the
// operands come straight from global memory / L2, two waves of a workgroup sharing each row panel through L1.
template <int EPI, int FORM, int TW>
__global__ __launch_bounds__(256) void k_gemm_q8(SkwGemmArgs a, SkwQ8Args qa) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * (32 * TW) + (w >> 1) * (16 * TW), n0 = blockIdx.x * (32 * TW) + (w & 1) * (16 * TW);
    const int nb = a.K >> 5;
    // operand rows (clamped: rows past the edge compute on a copy and are never stored)
    const int8_t* wp[TW]; const int8_t* ap[TW]; int mrow[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        wp[t] = qa.qw + (long)min(n0 + t * 16 + r16, a.N - 1) * a.K + g * 8;
        mrow[t] = min(m0 + t * 16 + r16, a.M - 1);
        ap[t] = qa.qa + (long)mrow[t] * a.K + g * 8;
    }
    float sumf[TW][TW][4];                                // [n tile][m tile][feature 4g + r]
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sumf[i][j][r] = 0.0f;
    // (requesting block b + 1's operands before block b is computed was measured slower: the second register set costs the second wave per SIMD)
    for (int b = 0; b < nb; ++b) {
        long fw[TW], fa[TW]; f32x4 dw[TW], mw[TW]; float dy[TW], sy[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            fw[t] = *(const long*)(wp[t] + b * 32); fa[t] = *(const long*)(ap[t] + b * 32);
            const int nq = min(n0 + t * 16 + 4 * g, qa.n_pad - 4);
            dw[t] = *(const f32x4*)(qa.dwT + (long)b * qa.n_pad + nq);
            if (FORM == 3) mw[t] = *(const f32x4*)(qa.mwT + (long)b * qa.n_pad + nq);
            dy[t] = qa.dyT[(long)b * a.M + mrow[t]];
            if (FORM == 3) sy[t] = qa.syT[(long)b * a.M + mrow[t]];
        }
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const i32x4 si = __builtin_amdgcn_mfma_i32_16x16x32_i8(fw[i], fa[j], (i32x4){0, 0, 0, 0}, 0, 0, 0);     // D[n = 4g + r][m = r16]
                q8_update4<FORM>(sumf[i][j], si, dw[i], FORM == 3 ? mw[i] : (f32x4){0.f, 0.f, 0.f, 0.f}, dy[j], FORM == 3 ? sy[j] : 0.0f);
            }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int m = m0 + j * 16 + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int n = n0 + i * 16 + 4 * g + r; if (n < a.N) q8_store<EPI>(a, m, n, sumf[i][j][r]); }
        }
}
// The encoder's shape, staged through LDS.  Same tile (128 x 128 per workgroup, four waves of 64 x 64) and the same arithmetic as k_gemm_q8<.., 4>,
// but the operands of two 32-blocks at a time — 64 bytes of every A and W row of the tile and the four scale / offset rows — arrive by LDS-DMA
// (global_load_lds, 16 B per lane, no registers in between) into one of two stage buffers while the other is being consumed, so a wave never
// waits on HBM / L2 with its 64 running sums idle.  A stage is [A 128 x 64 B][W 128 x 64 B][dw 2 x 128][mw 2 x 128][dy 2 x 128][sy 2 x 128] = 20 KB;
// the 16-byte pieces of a row are stored at piece ^ ((row >> 2) & 3) (by permuting which global piece a lane fetches), which spreads the sixteen
// rows of an 8-byte fragment read over all banks.  Requires N % 128 == 0, K % 64 == 0, M % 4 == 0 (every Whisper encoder product).
typedef __attribute__((address_space(1))) const void* q8_gptr_t;
typedef __attribute__((address_space(3))) void* q8_lptr_t;
template <int EPI, int FORM>
__global__ __launch_bounds__(256) void k_gemm_q8_lds(SkwGemmArgs a, SkwQ8Args qa) {
    constexpr int STAGE = 128 * 64 * 2 + 4 * 2 * 128 * 4;                    // 20480 bytes
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, wm = w >> 1, wn = w & 1;
    const int nbg = a.K >> 6;                                                 // stages: two 32-blocks each
    // DMA sources.  Operand tiles: DMA instruction q (0..7; wave w issues q = 2w, 2w + 1) covers tile rows 16q .. 16q + 15: lane -> row 16q + (lane >> 2),
    // LDS piece lane & 3, which must hold global piece (lane & 3) ^ ((row >> 2) & 3).
    const int8_t* srcA[2]; const int8_t* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * w + i) * 16 + (lane >> 2), piece = (lane & 3) ^ ((row >> 2) & 3);
        srcA[i] = qa.qa + (long)min(m0 + row, a.M - 1) * a.K + piece * 16;
        srcW[i] = qa.qw + (long)min(n0 + row, a.N - 1) * a.K + piece * 16;
    }
    // scale tables: wave 0 -> dw, 1 -> mw, 2 -> dy, 3 -> sy; lanes 0-31 fetch the first block's 128 values, lanes 32-63 the second block's
    const int sc_blk = lane >> 5, sc_col = (lane & 31) * 4;
    const float* sc_src; long sc_stride;
    if (w < 2) { sc_src = (w == 0 ? qa.dwT : qa.mwT) + n0 + sc_col; sc_stride = qa.n_pad; }
    else { sc_src = (w == 2 ? qa.dyT : qa.syT) + min(m0 + sc_col, a.M - 4); sc_stride = a.M; }
    auto stage = [&](int buf, int bg) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((q8_gptr_t)(srcA[i] + bg * 64), (q8_lptr_t)(base + (2 * w + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((q8_gptr_t)(srcW[i] + bg * 64), (q8_lptr_t)(base + 8192 + (2 * w + i) * 1024), 16, 0, 0);
        }
        if (FORM == 3 || (w != 1 && w != 3))
            __builtin_amdgcn_global_load_lds((q8_gptr_t)(sc_src + (long)(2 * bg + sc_blk) * sc_stride), (q8_lptr_t)(base + 16384 + w * 1024), 16, 0, 0);
    };
    // fragment offsets inside a stage: row * 64 + ((piece ^ ((row >> 2) & 3)) << 4) + (g & 1) * 8, piece = 2 * blk + (g >> 1)
    int offA[4], offW[4], colA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + r16, rw = wn * 64 + t * 16 + r16;
        offA[t] = ra * 64 + (g & 1) * 8; offW[t] = 8192 + rw * 64 + (g & 1) * 8; colA[t] = ra;
    }
    const int swA = (r16 >> 2) & 3;                                           // ((row >> 2) & 3) is the same for every 16-row tile of the wave: rows t * 16 + r16 with 16 | 16 t
    float sumf[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sumf[i][j][r] = 0.0f;
    stage(0, 0);
    for (int bg = 0; bg < nbg; ++bg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (bg + 1 < nbg) stage((bg + 1) & 1, bg + 1);
        const char* base = lds + (bg & 1) * STAGE;
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            long fw[4], fa[4]; f32x4 dw[4], mw[4]; float dy[4], sy[4];
            const int pc = (((2 * bl + (g >> 1)) ^ swA) << 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *(const long*)(base + offA[t] + pc); fw[t] = *(const long*)(base + offW[t] + pc);
                dw[t] = *(const f32x4*)(base + 16384 + (bl * 128 + wn * 64 + t * 16 + 4 * g) * 4);
                if (FORM == 3) mw[t] = *(const f32x4*)(base + 16384 + 1024 + (bl * 128 + wn * 64 + t * 16 + 4 * g) * 4);
                dy[t] = *(const float*)(base + 16384 + 2048 + (bl * 128 + colA[t]) * 4);
                if (FORM == 3) sy[t] = *(const float*)(base + 16384 + 3072 + (bl * 128 + colA[t]) * 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4 si = __builtin_amdgcn_mfma_i32_16x16x32_i8(fw[i], fa[j], (i32x4){0, 0, 0, 0}, 0, 0, 0);     // D[n = 4g + r][m = r16]
                    q8_update4<FORM>(sumf[i][j], si, dw[i], FORM == 3 ? mw[i] : (f32x4){0.f, 0.f, 0.f, 0.f}, dy[j], FORM == 3 ? sy[j] : 0.0f);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int n = n0 + wn * 64 + i * 16 + 4 * g + r; if (n < a.N) q8_store<EPI>(a, m, n, sumf[i][j][r]); }
        }
}

// The decode step's shape (M <= 128 rows).  A workgroup is one 16-feature strip x one 16-row tile; the block sum is cut into FOUR contiguous
// runs of blocks (the decoder's segmented contraction, D3' / D4: oracle linear_q8_seg), wave s chains run s block-ascending from zero with all of
// its operands requested up front (six blocks at K = 768), the partial sums meet in LDS and wave 0 adds them ((s0 + s1) + s2) + s3.
template <int EPI, int FORM, int UB>
__global__ __launch_bounds__(256) void k_gemm_q8_small(SkwGemmArgs a, SkwQ8Args qa) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int nb = a.K >> 5, bps = nb >> 2, b_lo = w * bps, mrow = min(m0 + r16, a.M - 1);      // host guarantees K % 128 == 0
    const int8_t* wp = qa.qw + (long)min(n0 + r16, a.N - 1) * a.K + g * 8;
    const int8_t* ap = qa.qa + (long)mrow * a.K + g * 8;
    const int nq = min(n0 + 4 * g, qa.n_pad - 4);
    float sumf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int b0 = 0; b0 < bps; b0 += UB) {
        long fw[UB], fa[UB]; f32x4 dw[UB], mw[UB]; float dy[UB], sy[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = b_lo + min(b0 + u, bps - 1);
            fw[u] = *(const long*)(wp + b * 32); fa[u] = *(const long*)(ap + b * 32);
            dw[u] = *(const f32x4*)(qa.dwT + (long)b * qa.n_pad + nq);
            if (FORM == 3) mw[u] = *(const f32x4*)(qa.mwT + (long)b * qa.n_pad + nq);
            dy[u] = qa.dyT[(long)b * a.M + mrow];
            if (FORM == 3) sy[u] = qa.syT[(long)b * a.M + mrow];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (b0 + u >= bps) break;
            const i32x4 si = __builtin_amdgcn_mfma_i32_16x16x32_i8(fw[u], fa[u], (i32x4){0, 0, 0, 0}, 0, 0, 0);
            q8_update4<FORM>(sumf, si, dw[u], FORM == 3 ? mw[u] : (f32x4){0.f, 0.f, 0.f, 0.f}, dy[u], FORM == 3 ? sy[u] : 0.0f);
        }
    }
    red[w][lane] = (f32x4){sumf[0], sumf[1], sumf[2], sumf[3]};
    __syncthreads();
    if (w != 0) return;
    const f32x4 s1 = red[1][lane], s2 = red[2][lane], s3 = red[3][lane];
    const int m = m0 + r16;
    if (m >= a.M) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = sumf[r] + s1[r]; v = v + s2[r]; v = v + s3[r];
        const int n = n0 + 4 * g + r; if (n < a.N) q8_store<EPI>(a, m, n, v);
    }
}
template <int EPI, int FORM> static void launch_gemm_q8_small(const SkwGemmArgs& a, const SkwQ8Args& qa, hipStream_t s) {
    const dim3 gs((a.N + 15) / 16, (a.M + 15) / 16); const int bps = (a.K >> 5) >> 2;
    if (bps <= 6) hipLaunchKernelGGL((k_gemm_q8_small<EPI, FORM, 6>), gs, dim3(256), 0, s, a, qa);
    else hipLaunchKernelGGL((k_gemm_q8_small<EPI, FORM, 8>), gs, dim3(256), 0, s, a, qa);
}
template <int EPI> static void launch_gemm_q8(const SkwGemmArgs& a, const SkwQ8Args& qa, hipStream_t s) {
    if (qa.segmented) {                     // the decoder's products
        if (qa.form == 1) launch_gemm_q8_small<EPI, 1>(a, qa, s); else if (qa.form == 2) launch_gemm_q8_small<EPI, 2>(a, qa, s); else launch_gemm_q8_small<EPI, 3>(a, qa, s);
        return;
    }
    if (skw_sw(SW_Q8_LDS) && a.M >= 1024 && !(a.N & 127) && !(a.K & 63) && !(a.M & 3)) {
        const dim3 grid(a.N / 128, (a.M + 127) / 128);
        if (qa.form == 1) hipLaunchKernelGGL((k_gemm_q8_lds<EPI, 1>), grid, dim3(256), 0, s, a, qa);
        else if (qa.form == 2) hipLaunchKernelGGL((k_gemm_q8_lds<EPI, 2>), grid, dim3(256), 0, s, a, qa);
        else hipLaunchKernelGGL((k_gemm_q8_lds<EPI, 3>), grid, dim3(256), 0, s, a, qa);
        return;
    }
    if (a.M >= 1024) {
        const dim3 grid((a.N + 127) / 128, (a.M + 127) / 128);
        if (qa.form == 1) hipLaunchKernelGGL((k_gemm_q8<EPI, 1, 4>), grid, dim3(256), 0, s, a, qa);
        else if (qa.form == 2) hipLaunchKernelGGL((k_gemm_q8<EPI, 2, 4>), grid, dim3(256), 0, s, a, qa);
        else hipLaunchKernelGGL((k_gemm_q8<EPI, 3, 4>), grid, dim3(256), 0, s, a, qa);
        return;
    }
    const dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
    if (qa.form == 1) hipLaunchKernelGGL((k_gemm_q8<EPI, 1, 2>), grid, dim3(256), 0, s, a, qa);
    else if (qa.form == 2) hipLaunchKernelGGL((k_gemm_q8<EPI, 2, 2>), grid, dim3(256), 0, s, a, qa);
    else hipLaunchKernelGGL((k_gemm_q8<EPI, 3, 2>), grid, dim3(256), 0, s, a, qa);
}
bool skw_gemm_q8(const SkwGemmArgs& a, const SkwQ8Args& qa, hipStream_t s) {
    if ((a.K & 31) || qa.form < 1 || qa.form > 3 || (qa.segmented && ((a.K & 127) || a.M > 4096))) return false;
    switch (a.epi) {
        case EPI_F32: launch_gemm_q8<EPI_F32>(a, qa, s); return true;
        case EPI_GELU_F32: launch_gemm_q8<EPI_GELU_F32>(a, qa, s); return true;
        case EPI_HEADS_F16: launch_gemm_q8<EPI_HEADS_F16>(a, qa, s); return true;
        case EPI_VT_F16: launch_gemm_q8<EPI_VT_F16>(a, qa, s); return true;
        case EPI_F16_PLAIN: launch_gemm_q8<EPI_F16_PLAIN>(a, qa, s); return true;
        case EPI_DEC_QKV: launch_gemm_q8<EPI_DEC_QKV>(a, qa, s); return true;
        default: return false;
    }
}

// token + position embedding from the dequantised (f32, not f16-rounded) token embedding: ggml_get_rows on a quantised tensor
__global__ void k_dec_embed_f32(const float* te32, const float* pe, const int* tok, const int* pos, int d, float* x) {
    const int b = blockIdx.x; const int tk = tok[b * (int)(sizeof(SkwSeqState) / 4)]; const int ps = pos[b * (int)(sizeof(SkwSeqState) / 4)];
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(long)b * d + i] = te32[(long)tk * d + i] + pe[(long)ps * d + i];
}
void skw_dec_embed_f32(const float* te32, const float* pe, const int* tok, const int* pos, int B, int d, float* x, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_embed_f32, dim3(B), dim3(256), 0, s, te32, pe, tok, pos, d, x);
}
