// Probe: shader clock seen by short dependent kernels (decode-like launch pattern) vs a long busy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_chain(float* out, unsigned long long* stamps, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float acc = out[threadIdx.x];
    _Float16 hm = (_Float16)out[threadIdx.x + 64], hw = (_Float16)1.0f;
    for (int i = 0; i < iters; i += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#ifdef MIX
            acc = __builtin_fmaf((float)hm, (float)hw, acc); asm volatile("" : "+v"(hm));
#else
            acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
#endif
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
int main() {
    float* d; unsigned long long* st; hipMalloc(&d, 4096); hipMalloc(&st, 16 * 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 3; ++mode) {
        int iters = mode == 0 ? 768 : mode == 1 ? 768 : 2000000; int blocks = mode == 2 ? 1024 : 48; int launches = mode == 2 ? 3 : 2000;
        std::vector<unsigned long long> h(2 * launches);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s);
        for (int i = 0; i < launches; ++i) { hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(mode == 1 ? 1024 : 64), 0, s, d, st + 2 * i, iters); }
        hipEventRecord(b, s); hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), st, 16 * launches, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0; for (int i = launches / 2; i < launches; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
        printf("mode %d: %d launches of %d-iter chains: %.3f ms total, %.2f us/launch; in-kernel: %.0f cycles, %.2f us -> clock %.0f MHz, %.2f cycles per dependent fma\n",
               mode, launches, iters, ms, 1000.0 * ms / launches, cyc / (launches / 2), rt / (launches / 2) / 100.0, cyc / rt * 100.0, cyc / (launches / 2) / iters);
    }
    return 0;
}
