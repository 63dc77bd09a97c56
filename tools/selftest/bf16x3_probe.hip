// Probe (not part of the library): accuracy of an fp32 product formed from three-way bf16 splits on
// v_mfma_f32_32x32x16_bf16 (6 or 9 partial products) against the exact-fp32 v_mfma_f32_32x32x2_f32 chain
// and an fp64 host reference.   hipcc --offload-arch=gfx950 -O3 -o bf16x3_probe.bin bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = x[j];
        const __bf16 bh = (__bf16)v;
        const float r1 = v - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const float r2 = r1 - (float)bm;
        h[j] = bh, m[j] = bm, l[j] = (__bf16)r2;
    }
}

// mode 0: fp32 MFMA; 6 / 9: bf16 splits.  One wave per 32 x 32 tile of C [M, N] = A [M, K] B [K, N].
__global__ void probe(const float* A, const float* B, float* C, int M, int N, int K, int mode) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long)(m0 + r) * K + k + h], B[(long)(k + h) * N + n0 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float a[8], b[8];
            for (int j = 0; j < 8; ++j) {
                a[j] = A[(long)(m0 + r) * K + k + 8 * h + j];
                b[j] = B[(long)(k + 8 * h + j) * N + n0 + r];
            }
            bf16x8 ah, am, al, bh, bm, bl;
            split3(a, ah, am, al);
            split3(b, bh, bm, bl);
            // smallest terms first
            if (mode == 9) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, acc, 0, 0, 0);
            }
            if (mode >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) C[(long)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + r] = acc[i];
}

// mode 16: low-order terms into a second accumulator, added at the end
__global__ void probe2(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    floatx16 acc, lo;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f, lo[i] = 0.f;
    for (int k = 0; k < K; k += 16) {
        float a[8], b[8];
        for (int j = 0; j < 8; ++j) {
            a[j] = A[(long)(m0 + r) * K + k + 8 * h + j];
            b[j] = B[(long)(k + 8 * h + j) * N + n0 + r];
        }
        bf16x8 ah, am, al, bh, bm, bl;
        split3(a, ah, am, al);
        split3(b, bh, bm, bl);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, lo, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[(long)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + r] = acc[i] + lo[i];
}

int main() {
    const int M = 64, N = 64;
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int K : {16, 64, 1024, 8192, 65536}) {
        std::vector<float> A((size_t)M * K), B((size_t)K * N), C((size_t)M * N);
        for (auto& v : A) v = nd(gen);
        for (auto& v : B) v = nd(gen);
        std::vector<double> ref((size_t)M * N, 0.0), mag((size_t)M * N, 0.0);
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < K; ++k) {
                const double a = A[(size_t)i * K + k];
                for (int j = 0; j < N; ++j) {
                    ref[(size_t)i * N + j] += a * B[(size_t)k * N + j];
                    mag[(size_t)i * N + j] += std::fabs(a * B[(size_t)k * N + j]);
                }
            }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int mode : {0, 3, 6, 9, 16}) {
            if (mode == 16)
                probe2<<<dim3(M / 32, N / 32), 64>>>(dA, dB, dC, M, N, K);
            else
                probe<<<dim3(M / 32, N / 32), 64>>>(dA, dB, dC, M, N, K, mode);
            if (hipDeviceSynchronize() != hipSuccess) return 1;
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double worst_mag = 0, worst_max = 0, refmax = 0, errmax = 0;
            for (size_t i = 0; i < C.size(); ++i) {
                const double e = std::fabs(C[i] - ref[i]);
                worst_mag = std::max(worst_mag, e / mag[i]);
                errmax = std::max(errmax, e);
                refmax = std::max(refmax, std::fabs(ref[i]));
            }
            worst_max = errmax / refmax;
            printf("K %6d mode %2d: max err / sum|ab| %.3e   max err / max|ref| %.3e\n", K, mode, worst_mag, worst_max);
        }
        hipFree(dA), hipFree(dB), hipFree(dC);
    }
    return 0;
}
