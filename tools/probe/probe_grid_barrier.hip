// Probe: cost of a device-wide barrier inside one persistent kernel (atomic counter + spin, agent-scope release/acquire fences so data
// written before the barrier by a workgroup on one XCD is visible after it to a workgroup on another) vs. a kernel boundary.
// Every wave has a bounded spin (bails out after 2^22 polls) so a logic error cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ bool grid_barrier(unsigned long long* counter, unsigned nblocks) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                           // release: this workgroup's stores (L2 write-back across XCDs)
        const unsigned long long old = atomicAdd(counter, 1ull);
        const unsigned long long target = (old / nblocks + 1) * nblocks;
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) { ok = false; break; } }
        __threadfence();                                           // acquire
    }
    __syncthreads();
    return ok;
}
__global__ void k_barriers(unsigned long long* counter, float* data, int n_barriers, int* fail) {
    const unsigned nb = gridDim.x;
    for (int i = 0; i < n_barriers; ++i) {
        // exchange: block b writes slot b, after the barrier reads its neighbour's slot written this round
        if (threadIdx.x == 0) data[blockIdx.x] = (float)(i * 1000 + blockIdx.x);
        if (!grid_barrier(counter, nb)) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
        if (threadIdx.x == 0) { const unsigned nbr = (blockIdx.x + 97) % nb; const float v = __builtin_nontemporal_load(&data[nbr]); if (v != (float)(i * 1000 + nbr)) atomicAdd(fail, 1000); }
        if (!grid_barrier(counter, nb)) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    }
}
__global__ void k_empty(float* data) { if (threadIdx.x == 0) data[blockIdx.x] += 1.0f; }
int main() {
    unsigned long long* counter; float* data; int* fail; hipMalloc(&counter, 8); hipMalloc(&data, 4096 * 4); hipMalloc(&fail, 4); hipMemset(counter, 0, 8); hipMemset(data, 0, 4096 * 4); hipMemset(fail, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int blocks : {48, 192, 256}) {
        const int nbar = 2000;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), 0, 0, counter, data, nbar, fail); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
        int hf = 0; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        printf("%3d workgroups: %.2f us per grid barrier (2 per round, %d rounds), failures %d\n", blocks, 1000.0 * ms / (2 * nbar), nbar, hf);
    }
    const int nl = 2000; hipEventRecord(e0, 0); for (int i = 0; i < nl; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, data); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("kernel boundary: %.2f us per dependent launch of a trivial kernel\n", 1000.0 * ms / nl);
    return 0;
}
