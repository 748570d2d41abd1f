// Runner for MFMA-semantics experiments (tools/probe/mfma_model.py generates the operands and fits a model to the results):
//   probe_mfma_run A.bin B.bin C.bin D.bin P     A: P x [16][32] f16 row-major, B: P x [32][16] f16 (k-major), C / D: P x [16][16] f32; one v_mfma_f32_16x16x32_f16 per problem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int p = blockIdx.x, l = threadIdx.x, r16 = l & 15, g = l >> 4;
    const _Float16* a = A + (size_t)p * 512; const _Float16* b = B + (size_t)p * 512; const float* c = C + (size_t)p * 256; float* d = D + (size_t)p * 256;
    f16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = a[r16 * 32 + 8 * g + e]; fb[e] = b[(8 * g + e) * 16 + r16]; }      // lane (row / col r16, group g) holds k = 8 g .. 8 g + 7
    f32x4 acc; for (int r = 0; r < 4; ++r) acc[r] = c[(4 * g + r) * 16 + r16];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + r16] = acc[r];
}
template <typename T> static std::vector<T> rd(const char* p, size_t n) { std::vector<T> v(n); FILE* f = fopen(p, "rb"); if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "read %s failed\n", p); exit(1); } fclose(f); return v; }
int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const size_t P = atol(argv[5]);
    auto A = rd<_Float16>(argv[1], P * 512); auto B = rd<_Float16>(argv[2], P * 512); auto C = rd<float>(argv[3], P * 256); std::vector<float> D(P * 256);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, P * 1024); hipMalloc(&dB, P * 1024); hipMalloc(&dC, P * 1024); hipMalloc(&dD, P * 1024);
    hipMemcpy(dA, A.data(), P * 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), P * 1024, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), P * 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(P), dim3(64), 0, 0, dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, P * 1024, hipMemcpyDeviceToHost);
    FILE* f = fopen(argv[4], "wb"); fwrite(D.data(), 4, D.size(), f); fclose(f);
    printf("ran %zu problems\n", P);
    return 0;
}
