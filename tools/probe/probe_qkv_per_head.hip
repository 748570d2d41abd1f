// Probe (round 5, VERDICT r4 item 4 i): what would the QKV half of a fused "QKV + self-attention per (head, 16-row tile)" decode kernel cost?
// The fusion puts one head's 192 weight rows (295 KB at d = 768) and 16 residual rows into ONE workgroup — 48 workgroups for 64 rows x 12 heads — where the product's
// k_gemm16_small_lnA spreads the same 3.5 MB over 144 workgroups (36 groups of four 16-column strips x 4 row tiles).  Both shapes are run here with the product kernel's structure
// (rows of x loaded as f32 and normalised in registers, four waves = four K quarters, weights from a fragment-order image straight into v_mfma_f32_16x16x32_f16, partial sums meeting
// in LDS), as chains of dependent launches over `cycle` copies of the weight (so that every launch finds its weights in HBM, as a decode step does):
//   shape A   36 x 4 workgroups, 4 strips each, one pass                      (the product's QKV launch)
//   shape B   12 x 4 workgroups, 12 strips each = q, k, v of one head, as three passes of four strips over the same normalised rows
//   shape C   12 x 8 workgroups (8-row tiles: half of every MFMA wasted), three passes     (twice the workgroups, twice the weight reads from L2)
//   shape D / E   the product shape with 8-row tiles (36 x 8) / with two strips per workgroup (72 x 4): more, lighter workgroups
// Prints microseconds per launch.  The fused kernel's attention half (reading 16 rows' K / V caches through the same CU) comes on top of B / C.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// K = 768 (6 k-blocks of 32 per wave), NT strips per pass, PASSES passes; rows_per_tile 16 or 8 (rows >= rows_per_tile of a tile load zeros)
template <int NT, int PASSES>
__global__ __launch_bounds__(256) void k_qkv(const float* x, const half_t* Wf, const float* gain, const float* bias, half_t* out, int M, int N, int K, int rows_per_tile, int strips_per_pass_stride) {
    constexpr int NKW = 6, NW = 4;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ f32x4 red[NW][NT][64];
    __shared__ f64x2 rowst[NW][16];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4, kb_lo = w * NKW, my0 = blockIdx.y * rows_per_tile;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wf, 0, (unsigned)((long)N * K * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)((long)M * K * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gain, 0, (unsigned)(K * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, (unsigned)(K * 4), 0x00020000);
    const unsigned oob = 0x7fffff00u, ko = (unsigned)((kb_lo * 32 + g * 8) * 4);
    const int m = my0 + r16; const unsigned xo = (r16 < rows_per_tile && m < M) ? (unsigned)((long)m * K * 4) + ko : oob;
    u32x4 fx[NKW][2], fw[NKW][NT];
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) fx[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo != oob ? xo + j * 128 + h * 16 : oob, 0, 0);
    auto wload = [&](int pass) {
#pragma unroll
        for (int j = 0; j < NKW; ++j)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int strip = pass * strips_per_pass_stride + blockIdx.x * NT + q;
                fw[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(((long)strip * (K >> 5) + kb_lo + j) * 1024 + lane * 16), 0, 0);
            }
    };
    wload(0);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) { const f32x4 v = __builtin_bit_cast(f32x4, fx[j][h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double xv = (double)v[e]; s1 += xv; s2 = __builtin_fma(xv, xv, s2); } }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (g == 0) rowst[w][r16] = (f64x2){s1, s2};
    u32x4 fg[NKW][2], fb[NKW][2];
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) { fg[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rg, ko + j * 128 + h * 16, 0, 0); fb[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rb, ko + j * 128 + h * 16, 0, 0); }
    __syncthreads();
    float sa, sb;
    { const f64x2 p0 = rowst[0][r16], p1 = rowst[1][r16], p2 = rowst[2][r16], p3 = rowst[3][r16];
      const double t1 = (p0[0] + p1[0]) + (p2[0] + p3[0]), t2 = (p0[1] + p1[1]) + (p2[1] + p3[1]);
      const double md = t1 / (double)K; double var = t2 / (double)K - md * md; if (var < 0.0) var = 0.0;
      const float rstd = 1.0f / sqrtf((float)var + 1e-5f); sa = rstd; sb = -(float)md * rstd; }
    f16x8 xa[NKW];
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
        const f32x4 g0 = __builtin_bit_cast(f32x4, fg[j][0]), g1 = __builtin_bit_cast(f32x4, fg[j][1]), b0 = __builtin_bit_cast(f32x4, fb[j][0]), b1 = __builtin_bit_cast(f32x4, fb[j][1]);
        const f32x4 x0 = __builtin_bit_cast(f32x4, fx[j][0]), x1 = __builtin_bit_cast(f32x4, fx[j][1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xa[j][e] = (half_t)__builtin_fmaf(__builtin_fmaf(x0[e], sa, sb), g0[e], b0[e]); xa[j][4 + e] = (half_t)__builtin_fmaf(__builtin_fmaf(x1[e], sa, sb), g1[e], b1[e]); }
    }
    for (int pass = 0; pass < PASSES; ++pass) {
        f32x4 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKW; ++j)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw[j][q]), xa[j], acc[q], 0, 0, 0);
        if (pass + 1 < PASSES) wload(pass + 1);                 // (the next pass's weights fly under this pass's reduction)
#pragma unroll
        for (int q = 0; q < NT; ++q) red[w][q][lane] = acc[q];
        __syncthreads();
        if (w < NT) {
            f32x4 v = red[0][w][lane];
#pragma unroll
            for (int s = 1; s < NW; ++s) { const f32x4 o = red[s][w][lane]; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            const int n = (pass * strips_per_pass_stride + blockIdx.x * NT + w) * 16 + 4 * g;
            if (r16 < rows_per_tile && m < M) { half_t* o = out + (long)m * N + n; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3]; }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int M = 64, K = 768, N = 2304, cycle = 96, iters = 480;      // 96 copies x 3.5 MB = 340 MB of weights walked per chain: past the 256 MB Infinity Cache, every launch streams from HBM
    float *x, *gain, *bias; half_t *W, *out;
    CHK(hipMalloc(&x, (size_t)M * K * 4)); CHK(hipMalloc(&gain, K * 4)); CHK(hipMalloc(&bias, K * 4)); CHK(hipMalloc(&W, (size_t)N * K * 2 * cycle)); CHK(hipMalloc(&out, (size_t)M * N * 2));
    { std::vector<float> h((size_t)M * K); unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; } CHK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CHK(hipMemcpy(gain, h.data(), K * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(bias, h.data() + K, K * 4, hipMemcpyHostToDevice));
      std::vector<unsigned short> hw((size_t)N * K); for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2c00 | ((s >> 9) & 0x3ff)); }
      for (int c = 0; c < cycle; ++c) CHK(hipMemcpy(W + (size_t)c * N * K, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
    hipStream_t st; CHK(hipStreamCreate(&st)); hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 24; ++i) launch(W + (size_t)(i % cycle) * N * K);
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch(W + (size_t)(i % cycle) * N * K);
        CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st)); CHK(hipGetLastError());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); printf("%-78s %7.2f us per dependent launch\n", name, 1e3 * ms / iters);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("A  product shape: 36 x 4 workgroups, 4 strips, one pass", [&](const half_t* w) { hipLaunchKernelGGL((k_qkv<4, 1>), dim3(36, 4), dim3(256), 0, st, x, w, gain, bias, out, M, N, K, 16, 0); });
        run("B  per head: 12 x 4 workgroups, q k v of one head as 3 passes of 4 strips", [&](const half_t* w) { hipLaunchKernelGGL((k_qkv<4, 3>), dim3(12, 4), dim3(256), 0, st, x, w, gain, bias, out, M, N, K, 16, 48); });
        run("D  product shape, 8-row tiles: 36 x 8 workgroups, 4 strips, one pass", [&](const half_t* w) { hipLaunchKernelGGL((k_qkv<4, 1>), dim3(36, 8), dim3(256), 0, st, x, w, gain, bias, out, M, N, K, 8, 0); });
        run("E  product shape, 2 strips: 72 x 4 workgroups, one pass", [&](const half_t* w) { hipLaunchKernelGGL((k_qkv<2, 1>), dim3(72, 4), dim3(256), 0, st, x, w, gain, bias, out, M, N, K, 16, 0); });
        run("C  per head, 8-row tiles: 12 x 8 workgroups, 3 passes", [&](const half_t* w) { hipLaunchKernelGGL((k_qkv<4, 3>), dim3(12, 8), dim3(256), 0, st, x, w, gain, bias, out, M, N, K, 8, 48); });
    }
    return 0;
}
