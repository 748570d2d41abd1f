// Probe: accumulation semantics of gfx950 MFMA (f16-in and f32-in forms).
// Dumps inputs + outputs; analysis is done offline (tools/probe/analyze_mfma.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// One wave per problem. A: [16][32] f16 row-major, B: [32][16] f16 row-major (k-major), C,D: [16][16] f32
__global__ void k_16x16x32_f16(const _Float16* A, const _Float16* B, const float* C, float* D) {
    int p = blockIdx.x; int l = threadIdx.x;
    const _Float16* a = A + (size_t)p * 16 * 32; const _Float16* b = B + (size_t)p * 32 * 16;
    const float* c = C + (size_t)p * 256; float* d = D + (size_t)p * 256;
    half8 fa, fb;
    for (int j = 0; j < 8; ++j) { int k = 8 * (l >> 4) + j; fa[j] = a[(l & 15) * 32 + k]; fb[j] = b[k * 16 + (l & 15)]; }
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[((l >> 4) * 4 + r) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// A: [32][16] f16, B: [16][32], C,D [32][32]
__global__ void k_32x32x16_f16(const _Float16* A, const _Float16* B, const float* C, float* D) {
    int p = blockIdx.x; int l = threadIdx.x;
    const _Float16* a = A + (size_t)p * 32 * 16; const _Float16* b = B + (size_t)p * 16 * 32;
    const float* c = C + (size_t)p * 1024; float* d = D + (size_t)p * 1024;
    half8 fa, fb;
    for (int j = 0; j < 8; ++j) { int k = 8 * (l >> 5) + j; fa[j] = a[(l & 31) * 16 + k]; fb[j] = b[k * 32 + (l & 31)]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); acc[r] = c[row * 32 + (l & 31)]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); d[row * 32 + (l & 31)] = acc[r]; }
}
// f32: 16x16x4: A [16][4], B [4][16]; chained NK times over K (K = 4*NK) A:[16][K], B:[K][16]
__global__ void k_16x16x4_f32(const float* A, const float* B, const float* C, float* D, int K) {
    int p = blockIdx.x; int l = threadIdx.x;
    const float* a = A + (size_t)p * 16 * K; const float* b = B + (size_t)p * K * 16;
    const float* c = C + (size_t)p * 256; float* d = D + (size_t)p * 256;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[((l >> 4) * 4 + r) * 16 + (l & 15)];
    for (int k0 = 0; k0 < K; k0 += 4) {
        int k = k0 + (l >> 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(l & 15) * K + k], b[k * 16 + (l & 15)], acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) d[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_32x32x2_f32(const float* A, const float* B, const float* C, float* D, int K) {
    int p = blockIdx.x; int l = threadIdx.x;
    const float* a = A + (size_t)p * 32 * K; const float* b = B + (size_t)p * K * 32;
    const float* c = C + (size_t)p * 1024; float* d = D + (size_t)p * 1024;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); acc[r] = c[row * 32 + (l & 31)]; }
    for (int k0 = 0; k0 < K; k0 += 2) {
        int k = k0 + (l >> 5);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(l & 31) * K + k], b[k * 32 + (l & 31)], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); d[row * 32 + (l & 31)] = acc[r]; }
}

static uint64_t rng_state = 0x1234567ULL;
static uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ULL; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static float rnd_f16val(int emin, int emax) { // random value exactly representable in f16 with exponent in [emin,emax]
    int e = emin + (int)(rnd() % (uint64_t)(emax - emin + 1)); int m = (int)(rnd() & 1023); int s = (int)(rnd() & 1);
    float v = ldexpf(1.0f + m / 1024.0f, e); return s ? -v : v;
}
template <typename T> static void dump(const char* name, const std::vector<T>& v) {
    char path[256]; snprintf(path, sizeof path, "gpurun_out/probe_%s.bin", name); FILE* f = fopen(path, "wb"); fwrite(v.data(), sizeof(T), v.size(), f); fclose(f);
}
template <typename T> static T* up(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch=%s CUs=%d clock=%d MHz mem=%.1f GB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1e9);
    const int P = 256;
    // ---- random tests, restricted exponents so exact sums fit in double ----
    {
        std::vector<_Float16> A(P * 16 * 32), B(P * 32 * 16); std::vector<float> C(P * 256), D(P * 256);
        for (auto& x : A) x = (_Float16)rnd_f16val(-3, 3);
        for (auto& x : B) x = (_Float16)rnd_f16val(-3, 3);
        for (auto& x : C) x = rnd_f16val(-4, 6) * (1.0f + (float)(rnd() & 8191) / 8388608.0f);
        auto dA = up(A); auto dB = up(B); auto dC = up(C); float* dD; CK(hipMalloc(&dD, D.size() * 4));
        k_16x16x32_f16<<<P, 64>>>(dA, dB, dC, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        dump("h16_A", A); dump("h16_B", B); dump("h16_C", C); dump("h16_D", D);
    }
    {
        std::vector<_Float16> A(P * 32 * 16), B(P * 16 * 32); std::vector<float> C(P * 1024), D(P * 1024);
        for (auto& x : A) x = (_Float16)rnd_f16val(-3, 3);
        for (auto& x : B) x = (_Float16)rnd_f16val(-3, 3);
        for (auto& x : C) x = rnd_f16val(-4, 6) * (1.0f + (float)(rnd() & 8191) / 8388608.0f);
        auto dA = up(A); auto dB = up(B); auto dC = up(C); float* dD; CK(hipMalloc(&dD, D.size() * 4));
        k_32x32x16_f16<<<P, 64>>>(dA, dB, dC, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        dump("h32_A", A); dump("h32_B", B); dump("h32_C", C); dump("h32_D", D);
    }
    // ---- ordering tests (16x16x32 f16): row r of problem p: products +2^24*.. pattern (i,j,m) ----
    {
        // triple index t = p*16 + row ; i = t/(32*32), j = (t/32)%32, m = t%32 ; p_i = +2^12*2^13=2^25, p_j = 1, p_m = -2^25 (if distinct)
        const int T = 32 * 32 * 32; const int PP = T / 16;
        std::vector<_Float16> A(PP * 16 * 32, (_Float16)0.0f), B(PP * 32 * 16, (_Float16)1.0f); std::vector<float> C(PP * 256, 0.0f), D(PP * 256);
        for (int t = 0; t < T; ++t) {
            int i = t / 1024, j = (t / 32) % 32, m = t % 32; int p = t / 16, row = t % 16;
            if (i == j || j == m || i == m) continue;
            A[(size_t)p * 512 + row * 32 + i] = (_Float16)4096.0f;   // times B=1 -> we need 2^25: use B scale below
            A[(size_t)p * 512 + row * 32 + j] = (_Float16)(1.0f / 8192.0f);
            A[(size_t)p * 512 + row * 32 + m] = (_Float16)(-4096.0f);
        }
        for (auto& x : B) x = (_Float16)8192.0f;   // products: +2^25, 1, -2^25
        auto dA = up(A); auto dB = up(B); auto dC = up(C); float* dD; CK(hipMalloc(&dD, D.size() * 4));
        k_16x16x32_f16<<<PP, 64>>>(dA, dB, dC, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        dump("ord16_D", D);
        // same but with C = +2^25 and products (j:1, m:-2^25): tests where C enters
        std::vector<float> C2(PP * 256, 33554432.0f), D2(PP * 256);
        std::vector<_Float16> A2(PP * 16 * 32, (_Float16)0.0f);
        for (int t = 0; t < 32 * 32; ++t) { int j = t / 32, m = t % 32; int p = t / 16, row = t % 16; if (j == m) continue;
            A2[(size_t)p * 512 + row * 32 + j] = (_Float16)(1.0f / 8192.0f); A2[(size_t)p * 512 + row * 32 + m] = (_Float16)(-4096.0f); }
        auto dA2 = up(A2); auto dC2 = up(C2);
        k_16x16x32_f16<<<64, 64>>>(dA2, dB, dC2, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D2.data(), dD, 64 * 256 * 4, hipMemcpyDeviceToHost)); D2.resize(64 * 256);
        dump("ordc16_D", D2);
    }
    // ---- f32 MFMA chain tests, K=64 ----
    for (int which = 0; which < 2; ++which) {
        const int K = 64; const int MN = which ? 32 : 16; const int PP = 1024;
        std::vector<float> A((size_t)PP * MN * K), B((size_t)PP * K * MN), C((size_t)PP * MN * MN), D((size_t)PP * MN * MN);
        for (auto& x : A) x = rnd_f16val(-3, 3) * (1.0f + (float)(rnd() & 8191) / 8388608.0f);
        for (auto& x : B) x = rnd_f16val(-3, 3) * (1.0f + (float)(rnd() & 8191) / 8388608.0f);
        for (auto& x : C) x = rnd_f16val(-3, 3);
        auto dA = up(A); auto dB = up(B); auto dC = up(C); float* dD; CK(hipMalloc(&dD, D.size() * 4));
        if (which) k_32x32x2_f32<<<PP, 64>>>(dA, dB, dC, dD, K); else k_16x16x4_f32<<<PP, 64>>>(dA, dB, dC, dD, K);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        dump(which ? "f32b_A" : "f32a_A", A); dump(which ? "f32b_B" : "f32a_B", B); dump(which ? "f32b_C" : "f32a_C", C); dump(which ? "f32b_D" : "f32a_D", D);
    }
    printf("probe done\n");
    return 0;
}
