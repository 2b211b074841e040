// Feasibility probe for split-precision projections: fp32 product emulated by six bf16 MFMAs
// (operands split into three bf16 pieces h + m + l; products hh, hm, mh, hl, lh, mm; fp32 accumulation).
//  1. accuracy of C = A W^T (M x K x N = 4096 x 276 x 138) against fp64, next to the fp32-MFMA result
//  2. throughput of the NT-like step (A split in registers, W pieces read from LDS) with random operands
// Build: hipcc -O3 --offload-arch=gfx950 tools/bf16x6_lab.hip -o tools/bin/bf16x6_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// W pieces: [3][N16][Kp] bf16, N16 = 144, Kp = 288
__global__ void k_split_w(const float* W, int N, int K, __bf16* out, int N16, int Kp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N16 * Kp) return;
    const int n = e / Kp, k = e % Kp;
    const float x = (n < N && k < K) ? W[n * K + k] : 0.f;
    __bf16 h, m, l;
    split3(x, h, m, l);
    out[e] = h; out[N16 * Kp + e] = m; out[2 * N16 * Kp + e] = l;
}

// one wave = 16 rows x 144 columns; MODE 0: bf16x6, MODE 1: fp32 MFMA (reference accuracy)
template <int MODE>
__global__ __launch_bounds__(64) void k_gemm(const float* A, int lda, int K, const __bf16* Wp, const float* Wf, int N16, int Kp,
                                             float* C, int ldc) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    const int row = blockIdx.x * 16 + i;
    f32x4 acc[9];
    for (int n = 0; n < 9; ++n) acc[n] = f32x4{0, 0, 0, 0};
    if (MODE == 0) {
        for (int k0 = 0; k0 < Kp; k0 += 32) {
            bf16x8 ah, am, al;
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kq * 8 + j;
                const float x = k < K ? A[row * lda + k] : 0.f;
                __bf16 h, m, l;
                split3(x, h, m, l);
                ah[j] = h; am[j] = m; al[j] = l;
            }
            for (int n = 0; n < 9; ++n) {
                const __bf16* w = Wp + (size_t)(16 * n + i) * Kp + k0 + kq * 8;
                const bf16x8 wh = *(const bf16x8*)w, wm = *(const bf16x8*)(w + (size_t)N16 * Kp), wl = *(const bf16x8*)(w + (size_t)2 * N16 * Kp);
                // small terms first
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, acc[n], 0, 0, 0);
            }
        }
    } else {
        for (int k0 = 0; k0 < Kp; k0 += 4) {
            const int k = k0 + kq;
            const float a = k < K ? A[row * lda + k] : 0.f;
            for (int n = 0; n < 9; ++n) {
                const int nn = 16 * n + i;
                const float w = (k < K) ? Wf[nn * Kp + k] : 0.f;
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, a, acc[n], 0, 0, 0);
            }
        }
    }
    for (int n = 0; n < 9; ++n)
        for (int r = 0; r < 4; ++r) C[(size_t)row * ldc + 16 * n + 4 * kq + r] = acc[n][r];
}

// throughput: NT-like step with W pieces in LDS (3 x 144 x 32 bf16 per 32-wide k chunk, cycled), A split in registers
__global__ __launch_bounds__(1024) void k_rate(float* out, int iters, const float* Aseed, const __bf16* Wp) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];      // [4 chunks][3][144][32]
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    for (int e = t; e < 4 * 3 * 144 * 32; e += blockDim.x) lds[e] = Wp[e % (3 * 144 * 288)];
    __syncthreads();
    f32x4 acc[9];
    for (int n = 0; n < 9; ++n) acc[n] = f32x4{0, 0, 0, 0};
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = Aseed[(blockIdx.x * 1024 + t) * 8 + j];
    for (int it = 0; it < iters; ++it) {
        bf16x8 ah, am, al;
        for (int j = 0; j < 8; ++j) {
            __bf16 h, m, l;
            split3(a[j], h, m, l);
            ah[j] = h; am[j] = m; al[j] = l;
            a[j] = a[j] * 1.0001f + 1e-3f;          // fresh values every step, like streamed rows
        }
        const __bf16* base = lds + (size_t)(it & 3) * 3 * 144 * 32 + (size_t)i * 32 + kq * 8;
        for (int n = 0; n < 9; ++n) {
            const __bf16* w = base + (size_t)16 * n * 32;
            const bf16x8 wh = *(const bf16x8*)w, wm = *(const bf16x8*)(w + 144 * 32), wl = *(const bf16x8*)(w + 2 * 144 * 32);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, acc[n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < 9; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 1024 + t] = s;
}

int main() {
    const int M = 4096, K = 276, N = 138, N16 = 144, Kp = 288, lda = 280, ldc = 144;
    std::vector<float> hA((size_t)M * lda, 0.f), hW((size_t)N * K), hWf((size_t)N16 * Kp, 0.f);
    unsigned x = 123u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; };
    for (int r = 0; r < M; ++r) for (int k = 0; k < K; ++k) hA[(size_t)r * lda + k] = rnd();
    for (auto& v : hW) v = rnd() * 0.3f;
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) hWf[(size_t)n * Kp + k] = hW[(size_t)n * K + k];
    float *dA, *dW, *dWf, *dC; __bf16* dWp;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dW, hW.size() * 4); hipMalloc(&dWf, hWf.size() * 4);
    hipMalloc(&dC, (size_t)M * ldc * 4); hipMalloc(&dWp, (size_t)3 * N16 * Kp * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dWf, hWf.data(), hWf.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split_w, dim3((N16 * Kp + 255) / 256), dim3(256), 0, 0, dW, N, K, dWp, N16, Kp);
    std::vector<float> c0((size_t)M * ldc), c1((size_t)M * ldc);
    hipLaunchKernelGGL(k_gemm<0>, dim3(M / 16), dim3(64), 0, 0, dA, lda, K, dWp, dWf, N16, Kp, dC, ldc);
    hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_gemm<1>, dim3(M / 16), dim3(64), 0, 0, dA, lda, K, dWp, dWf, N16, Kp, dC, ldc);
    hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, mx = 0, s0 = 0, s1 = 0;
    for (int r = 0; r < M; r += 7)
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * lda + k] * (double)hW[(size_t)n * K + k];
            const double d0 = fabs(c0[(size_t)r * ldc + n] - ref), d1 = fabs(c1[(size_t)r * ldc + n] - ref);
            e0 = fmax(e0, d0); e1 = fmax(e1, d1); mx = fmax(mx, fabs(ref)); s0 += d0 * d0; s1 += d1 * d1;
        }
    printf("accuracy vs fp64 (max |C| = %.3f): bf16x6 max err %.3e rms %.3e | fp32 MFMA max err %.3e rms %.3e\n", mx, e0,
           sqrt(s0 / (M / 7 * N)), e1, sqrt(s1 / (M / 7 * N)));

    // throughput
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    float* seed; hipMalloc(&seed, (size_t)256 * 1024 * 8 * 4);
    std::vector<float> hs((size_t)256 * 1024 * 8);
    for (auto& v : hs) v = rnd();
    hipMemcpy(seed, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)4 * 3 * 144 * 32 * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int wpb : {4, 8, 16}) {
        hipEvent_t e0_, e1_; hipEventCreate(&e0_); hipEventCreate(&e1_);
        const int iters = 4000;
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(64 * wpb), ldsb, 0, out, iters / 10, seed, dWp);
        hipDeviceSynchronize();
        hipEventRecord(e0_);
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(64 * wpb), ldsb, 0, out, iters, seed, dWp);
        hipEventRecord(e1_); hipEventSynchronize(e1_);
        float ms; hipEventElapsedTime(&ms, e0_, e1_);
        const double flops_equiv = 2.0 * 16 * 144 * 32 * (double)iters * 256 * wpb;     // fp32-equivalent useful flops
        printf("bf16x6 NT step, %2d waves/CU: %.3f ms  %.1f TF/s fp32-equivalent (%.1f TF/s of bf16 MFMA)\n", wpb, ms,
               flops_equiv / ms / 1e9, 6 * flops_equiv / ms / 1e9);
    }
    return 0;
}
