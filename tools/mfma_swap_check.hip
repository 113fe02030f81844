// Is v_mfma_f32_16x16x4_f32 symmetric under swapping its operands?  D1 = A.B + C against D2 = B^T.A^T + C^T, bit for bit.
// (round 5: the register-resident lean chain computes every product transposed; the resident units emit transposed slabs by
//  swapping the MFMA operands.)  build: hipcc --offload-arch=gfx950 -O2 tools/mfma_swap_check.hip -o /tmp/mfma_swap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D1, float* D2, int chain) {
    const int lane = threadIdx.x, l15 = lane & 15, lg = lane >> 4;
    const int t = blockIdx.x;
    A += t * 64 * chain; B += t * 64 * chain; C += t * 256; D1 += t * 256; D2 += t * 256;
    f32x4 c1, c2;
    for (int q = 0; q < 4; ++q) { c1[q] = C[(4 * lg + q) * 16 + l15]; c2[q] = C[l15 * 16 + 4 * lg + q]; }   // c2 = C^T image
    for (int s = 0; s < chain; ++s) {
        const float a = A[s * 64 + l15 * 4 + lg];      // A[m = l15][k = lg]
        const float b = B[s * 64 + lg * 16 + l15];     // B[k = lg][n = l15]
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c2, 0, 0, 0);   // B^T[m = n][k] . A^T[k][n = m]
    }
    for (int q = 0; q < 4; ++q) { D1[(4 * lg + q) * 16 + l15] = c1[q]; D2[l15 * 16 + 4 * lg + q] = c2[q]; }   // both stored as D[m][n]
}
int main() {
    const int T = 4096, chain = 4;
    std::vector<float> A(T * 64 * chain), B(T * 64 * chain), C(T * 256), D1(T * 256), D2(T * 256);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / float(1 << 20); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd() * 1e-2f;
    for (auto& v : C) v = rnd() * 3.0f;
    float *dA, *dB, *dC, *d1, *d2;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&d1, D1.size() * 4); hipMalloc(&d2, D2.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    k<<<T, 64>>>(dA, dB, dC, d1, d2, chain);
    hipMemcpy(D1.data(), d1, D1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(D2.data(), d2, D2.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < D1.size(); ++i) bad += memcmp(&D1[i], &D2[i], 4) != 0;
    printf("mfma_f32_16x16x4_f32 operand swap: %zu of %zu elements differ (%s)\n", bad, D1.size(), bad ? "NOT symmetric" : "bit-identical transpose");
    return 0;
}
