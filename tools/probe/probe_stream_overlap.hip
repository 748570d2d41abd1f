// Probe: do two HIP streams, each carrying a chain of dependent small kernels (the shape of a decode step), overlap on the device?
// Each kernel: 64 workgroups, a short dependent pointer walk (~4-5 us).  Prints the wall time of one chain alone and of two chains
// on two streams, launched interleaved kernel by kernel and launched chain after chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_walk(const int* next, int* out, int hops) {
    int p = blockIdx.x * 64 + (threadIdx.x & 63);
    for (int i = 0; i < hops; ++i) p = __builtin_nontemporal_load(&next[p]);
    if (threadIdx.x == 0) out[blockIdx.x] = p;
}
int main() {
    const int N = 1 << 22; std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 1664525 + 1013904223) % N);
    int *next, *outA, *outB; hipMalloc(&next, N * 4); hipMalloc(&outA, 4096); hipMalloc(&outB, 4096); hipMemcpy(next, h.data(), N * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, eb; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb); float ms;
    const int n = 400, hops = 4;
    auto run = [&](int mode) {
        hipDeviceSynchronize();
        hipEventRecord(e0, sa); hipStreamWaitEvent(sb, e0, 0);
        if (mode == 0) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_walk, dim3(64), dim3(256), 0, sa, next, outA, hops);
        if (mode == 1) for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_walk, dim3(64), dim3(256), 0, sa, next, outA, hops); hipLaunchKernelGGL(k_walk, dim3(64), dim3(256), 0, sb, next, outB, hops); }
        if (mode == 2) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_walk, dim3(64), dim3(256), 0, sa, next, outA, hops); for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_walk, dim3(64), dim3(256), 0, sb, next, outB, hops); }
        hipEventRecord(eb, sb); hipStreamWaitEvent(sa, eb, 0); hipEventRecord(e1, sa); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run(0), t1 = run(1), t2 = run(2);
        printf("one chain of %d kernels: %.3f ms (%.2f us each); two chains interleaved: %.3f ms; two chains back to back on two streams: %.3f ms\n", n, t0, 1000 * t0 / n, t1, t2);
    }
    return 0;
}
