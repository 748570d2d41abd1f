// Probe: issue rate of v_mfma_f32_16x16x4_f32 over 16 independent accumulators (the k_gemm inner loop shape), with and without
// the 8 operand conversions per 16 MFMAs, at 1..4 waves per SIMD.  Whole-kernel timing, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union H8v { u32x4 v; _Float16 h[8]; };
template <int MODE>
__global__ void k_ind(float* out, const u32x4* in, int iters) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    H8v fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[t].v = in[threadIdx.x + 64 * t]; fb[t].v = in[threadIdx.x + 64 * (t + 4)]; }
    float av[4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { av[t] = out[threadIdx.x + t]; bv[t] = out[threadIdx.x + 8 + t]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { asm volatile("" : "+v"(fa[t].v), "+v"(fb[t].v)); av[t] = (float)fa[t].h[e]; bv[t] = (float)fb[t].h[e]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[threadIdx.x + 128] = s;
}
int main() {
    float* d; u32x4* in; hipMalloc(&d, 1 << 16); hipMalloc(&in, 1 << 16); hipMemset(d, 0, 1 << 16); hipMemset(in, 0, 1 << 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int waves = 1; waves <= 4; ++waves) {
        const int iters = 4000; float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k_ind<0>, dim3(256 * waves), dim3(256), 0, 0, d, in, iters);
            else hipLaunchKernelGGL(k_ind<1>, dim3(256 * waves), dim3(256), 0, 0, d, in, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double n_mfma = (double)iters * 128 * waves;   // per SIMD
        printf("mode %d (%s), %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD at 2.39 GHz -> %.1f%% of the 32-cycle rate\n", mode, mode ? "64 cvt per 128 MFMA" : "no VALU", waves, 2.39e6 * ms / n_mfma, 100.0 * 32.0 / (2.39e6 * ms / n_mfma));
    }
    return 0;
}
