// Probe: cost of a dependent v_mfma_f32_16x16x4_f32 chain when (a) operands are loop-invariant registers,
// (b) operands come from v_cvt_f32_f16 of packed registers (as in k_gemm_smallm), (c) as (b) with SDWA-free unpacking.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union H8v { u32x4 v; _Float16 h[8]; };
template <int MODE>
__global__ void k_chain(float* out, const u32x4* in, unsigned long long* stamps, int iters) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    H8v a, w; a.v = in[threadIdx.x]; w.v = in[threadIdx.x + 64];
    float fa = out[threadIdx.x], fw = out[threadIdx.x + 64];
    float pa[8], pw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { pa[e] = fa; pw[e] = fw; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i += 8) {
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fw, acc, 0, 0, 0);
        } else if (MODE == 1) {
            asm volatile("" : "+v"(a.v), "+v"(w.v));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a.h[e], (float)w.h[e], acc, 0, 0, 0);
        } else if (MODE == 2) {      // all 16 conversions first, then the 8 MFMAs
            asm volatile("" : "+v"(a.v), "+v"(w.v));
            float xa[8], xw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xa[e] = (float)a.h[e]; xw[e] = (float)w.h[e]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], xw[e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                     // conversions for the next 8 issued between this block's MFMAs (one pair after each MFMA)
            asm volatile("" : "+v"(a.v), "+v"(w.v));
            float xa[8], xw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xa[e] = pa[e]; xw[e] = pw[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], xw[e], acc, 0, 0, 0);
                pa[e] = (float)a.h[e]; pw[e] = (float)w.h[e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + 128] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
}
int main() {
    float* d; u32x4* in; unsigned long long* st; hipMalloc(&d, 4096); hipMalloc(&in, 4096); hipMalloc(&st, 64); hipMemset(d, 0, 4096); hipMemset(in, 0, 4096);
    for (int mode = 0; mode < 4; ++mode) for (int waves = 1; waves <= 2; ++waves) {
        const int iters = 768; unsigned long long h = 0; double cyc = 0;
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(48), dim3(256 * waves), 0, 0, d, in, st, iters);
            else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(48), dim3(256 * waves), 0, 0, d, in, st, iters);
            else if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(48), dim3(256 * waves), 0, 0, d, in, st, iters);
            else hipLaunchKernelGGL(k_chain<3>, dim3(48), dim3(256 * waves), 0, 0, d, in, st, iters);
            hipDeviceSynchronize(); hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); if (rep >= 10) cyc += h;
        }
        printf("mode %d (%s), %d wave(s)/SIMD: %.1f memtime cycles per dependent MFMA\n", mode, mode == 0 ? "invariant operands" : mode == 1 ? "cvt operands, compiler order" : mode == 2 ? "cvt block then mfma block" : "cvt for next block interleaved", waves, cyc / 10 / iters);
    }
    // whole-kernel timing (events), long chains: ns per MFMA per SIMD, all 256 CUs busy
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) for (int waves = 1; waves <= 3; ++waves) {
        const int iters = 200000; float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(256), dim3(256 * waves), 0, 0, d, in, st, iters);
            else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(256), dim3(256 * waves), 0, 0, d, in, st, iters);
            else if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(256), dim3(256 * waves), 0, 0, d, in, st, iters);
            else hipLaunchKernelGGL(k_chain<3>, dim3(256), dim3(256 * waves), 0, 0, d, in, st, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("events: mode %d, %d wave(s)/SIMD: %.2f ns per MFMA per SIMD (%.1f cycles at 2.39 GHz)\n", mode, waves, 1e6 * ms / iters / waves, 2.39 * 1e6 * ms / iters / waves);
    }
    return 0;
}
