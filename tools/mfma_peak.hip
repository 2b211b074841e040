// Peak-rate probe for v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32: register-only loops, no memory.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT> struct ResGroups {
    static constexpr int count = NT < 2 ? 1 : NT / 2;
    static constexpr int start(int g) { return 2 * g; }
    static constexpr int size(int g) { return g == count - 1 ? NT - 2 * g : 2; }
};
template <int NT, int RT, int G>
struct ResSched {
    static __device__ __forceinline__ void run() {
        using RG = ResGroups<NT>;
        if constexpr (G < RG::count) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RG::size(G) * RT, 0);
            if constexpr (G + 2 < RG::count) __builtin_amdgcn_sched_group_barrier(0x100, RG::size(G + 2), 0);
            ResSched<NT, RT, G + 1>::run();
        }
    }
};

// NT-like step: 9 ds_read_b128 weight quads + 36 MFMAs (a from a register), reads issued one pair ahead
template <int MODE, int NVALU = 0>
__global__ __launch_bounds__(1024) void knt(float* out, int iters, float a0, const float* w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    for (int e = t; e < 36 * 144 * 4; e += blockDim.x) lds[e] = w[e & 1023];
    __syncthreads();
    f32x4 acc[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = {a0 + lane, a0, 1.f, 2.f};
    for (int it = 0; it < iters; ++it) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(lds) + ((it & 7) * 4 + kq) * 144 + i;
        if (MODE == 0) {          // all reads up front
            f32x4 b[9];
#pragma unroll
            for (int n = 0; n < 9; ++n) b[n] = wp[16 * n];
#pragma unroll
            for (int n = 0; n < 9; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n][m], a[m], acc[n], 0, 0, 0);
        } else if (MODE == 1) {   // no LDS in the loop
            const f32x4 b = {a0, a0 + 1, a0 + 2, a0 + 3};
#pragma unroll
            for (int n = 0; n < 9; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[m], a[m], acc[n], 0, 0, 0);
        } else if (MODE == 3) {   // the production schedule: groups 2,2,2,3 with pinned order
            using RG = ResGroups<9>;
            f32x4 b[9];
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int n = RG::start(gq); n < RG::start(gq) + RG::size(gq); ++n) b[n] = wp[16 * n];
#pragma unroll
            for (int gq = 0; gq < RG::count; ++gq) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = RG::start(gq); n < RG::start(gq) + RG::size(gq); ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n][m], a[m], acc[n], 0, 0, 0);
                if (gq + 2 < RG::count) {
#pragma unroll
                    for (int n = RG::start(gq + 2); n < RG::start(gq + 2) + RG::size(gq + 2); ++n) b[n] = wp[16 * n];
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            ResSched<9, 1, 0>::run();
        } else {                  // pairs, interleaved m
            f32x4 b0 = wp[0], b1 = wp[16];
#pragma unroll
            for (int n = 0; n < 9; n += 2) {
                f32x4 n0 = b0, n1 = b1;
                if (n + 2 < 9) n0 = wp[16 * (n + 2)];
                if (n + 3 < 9) n1 = wp[16 * (n + 3)];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[m], a[m], acc[n], 0, 0, 0);
                    if (n + 1 < 9) acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[m], a[m], acc[n + 1], 0, 0, 0);
                }
                b0 = n0; b1 = n1;
            }
        }
        a[0] += 1.f;
        if (NVALU > 0) {
            float dv = a[1];
#pragma unroll
            for (int q = 0; q < NVALU; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(dv) : "v"(a0));
            a[1] = dv;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + t] = s;
}


// Lab kernel: the production NT step with features of the real kernel switched on one by one.
//  F & 1 : weights spread over 161 KB of LDS, address from a per-step table (LDS b64 read one step ahead)
//  F & 2 : A quads come from a 3-slot register ring filled by global loads (counted vmcnt), a-select by table bit
//  F & 4 : 18-step tiles: accumulators initialised from LDS and stored to global per tile
//  F & 8 : waves get 2 or 3 tiles (like 10062 tiles over 4096 waves) instead of 3 each
template <int F>
__global__ __launch_bounds__(1024) void klab(float* out, int tiles_per_wave, const float* A, int64_t lda, int64_t M, const float* w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int R = 144, NG = 70;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int e = t; e < NG * R * 4; e += blockDim.x) lds[e] = w[e & 1023];
    int2* tab = reinterpret_cast<int2*>(lds + NG * R * 4);
    for (int e = t; e < 18 * 4; e += blockDim.x) {
        const int fq = e >> 2, k = e & 3;
        const int q = fq % 9, gi = (q < 8) ? 8 * (q >> 1) + 2 * k + (q & 1) : 32 + k;
        const int gc = gi < 35 ? gi : 34;
        tab[e] = make_int2((gc * 16) | (gi < 35 ? 0 : (int)0x80000000), ((fq >= 9 ? 35 : 0) + gc) * R * 16);
    }
    __syncthreads();
    using RG = ResGroups<9>;
    f32x4 acc[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t gw = (int64_t)wib * gridDim.x + blockIdx.x, nw = (int64_t)16 * gridDim.x;
    int my_tiles = tiles_per_wave;
    if (F & 8) my_tiles = gw < 1870 ? 3 : 2;
    const int total = my_tiles * 18;
    f32x4 ring[3];
    int2 ring_e[3];
    int lstep = 0;
    const char* rowp = reinterpret_cast<const char*>(A + min(gw * 16 + i, M - 1) * lda);
    int lfq = 0; int64_t ltile = gw;
    auto issue = [&](int d) {
        const int fq = lfq;
        const int2 e = (F & 1) ? tab[fq * 4 + kq] : make_int2(0, ((lstep & 7) * 4 + kq) * R * 16);
        ring_e[d] = e;
        if (F & 2) ring[d] = *reinterpret_cast<const f32x4*>(rowp + (e.x & 0x3fffffff));
        else ring[d] = f32x4{1.f, 2.f, 3.f, (float)lstep};
        ++lstep;
        if (++lfq == 18) { lfq = 0; if (lstep < total) { ltile += nw; rowp = reinterpret_cast<const char*>(A + min(ltile * 16 + i, M - 1) * lda); } }
    };
#pragma unroll
    for (int d = 0; d < 3; ++d) issue(d);
    f32x4 bnext[4];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ring_e[0].y) + i;
#pragma unroll
        for (int n = 0; n < 4; ++n) bnext[n] = wp[16 * n];
    }
    for (int s0 = 0; s0 < total; s0 += 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const f32x4 ain = ring[d];
            const int2 e = ring_e[d];
            issue(d);
            const int2 ef = ring_e[(d + 1) % 3];
            const f32x4 a = ((F & 2) && e.x < 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : ain;
            const f32x4* wp = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + e.y) + i;
            const f32x4* wf = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ef.y) + i;
            f32x4 b[9];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = bnext[n];
#pragma unroll
            for (int gq = 0; gq < RG::count; ++gq) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = RG::start(gq); n < RG::start(gq) + RG::size(gq); ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n][m], a[m], acc[n], 0, 0, 0);
                if (gq + 2 < RG::count) {
#pragma unroll
                    for (int n = RG::start(gq + 2); n < RG::start(gq + 2) + RG::size(gq + 2); ++n) b[n] = wp[16 * n];
                }
                if (gq == RG::count - 2) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bnext[n] = wf[16 * n];
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            if ((F & 4) && (s0 + d) % 18 == 17) {
                const int64_t tile = gw + (int64_t)((s0 + d) / 18) * nw;
                float* c = out + ((tile * 16 + i) % M) * 144;
#pragma unroll
                for (int n = 0; n < 9; ++n) {
                    *reinterpret_cast<f32x4*>(c + 16 * n + 4 * kq) = acc[n];
                    acc[n] = *reinterpret_cast<const f32x4*>(&lds[16 * n + 4 * kq]);
                }
            }
        }
    }
    if (!(F & 4)) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 9; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        out[blockIdx.x * blockDim.x + t] = s;
    }
}

template <int F>
void run_lab(float* out, const float* A, const float* w) {
    const size_t ldsb = 70 * 144 * 16 + 18 * 4 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(klab<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int64_t M = 160984;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(klab<F>, dim3(256), dim3(1024), ldsb, 0, out, 3, A, (int64_t)140, M, w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(klab<F>, dim3(256), dim3(1024), ldsb, 0, out, 3, A, (int64_t)140, M, w);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tiles = (F & 8) ? 10062.0 : 4096.0 * 3;
    printf("lab F=%2d  %7.1f us per launch  %6.1f TF/s (MFMA level)\n", F, ms * 100.0, tiles * 648 * 2048.0 / (ms / 10 * 1e-3) / 1e12);
}

template <typename F>
void run(const char* name, F launch, double flops_per_wave_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = flops_per_wave_iter * iters * blocks * 4.0;
    printf("%-40s blocks=%5d  %8.3f ms  %7.1f TF/s\n", name, blocks, ms, fl / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {           // waves per CU = 4 * blocks per CU
        const int blocks = 256 * wpc / 4;
        char nm[64];
        snprintf(nm, sizeof nm, "16x16x4 f32, 9 acc, %d waves/CU", wpc);
        run(nm, [&](int it) { hipLaunchKernelGGL(k16<9>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 9 * 2048.0, blocks, iters);
        snprintf(nm, sizeof nm, "16x16x4 f32, 2 acc, %d waves/CU", wpc);
        run(nm, [&](int it) { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 2 * 2048.0, blocks, iters);
        snprintf(nm, sizeof nm, "32x32x2 f32, 4 acc, %d waves/CU", wpc);
        run(nm, [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4 * 4096.0, blocks, iters);
    }
    float* w;
    hipMalloc(&w, 4096);
    hipMemset(w, 0, 4096);
    float* A;
    hipMalloc(&A, (size_t)160984 * 140 * 4);
    hipMemset(A, 0, (size_t)160984 * 140 * 4);
    float* C;
    hipMalloc(&C, (size_t)160984 * 144 * 4);
    run_lab<15>(C, A, w);
    {   // same again with random operands instead of zeros (switching activity -> power -> clock)
        const size_t na = (size_t)160984 * 140;
        float* h = (float*)malloc(na * 4);
        unsigned x = 12345u;
        for (size_t q = 0; q < na; ++q) { x = x * 1664525u + 1013904223u; h[q] = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
        hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, h, 4096, hipMemcpyHostToDevice);
        free(h);
        printf("random operands:\n");
        run_lab<15>(C, A, w); run_lab<0>(C, A, w);
    }
    run_lab<0>(C, A, w); run_lab<1>(C, A, w); run_lab<3>(C, A, w); run_lab<7>(C, A, w); run_lab<15>(C, A, w);
    run_lab<4>(C, A, w); run_lab<2>(C, A, w); run_lab<8>(C, A, w);
    const size_t ldsb = 36 * 144 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<3, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<3, 20>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(knt<3, 40>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int wpb : {4, 16}) {
        char nm2[64];
        auto fl2 = 36 * 2048.0 * wpb / 4.0;
        snprintf(nm2, sizeof nm2, "NT step + 10 VALU, %d waves/CU", wpb);
        run(nm2, [&](int it) { hipLaunchKernelGGL((knt<3, 10>), dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl2, 256, 4000);
        snprintf(nm2, sizeof nm2, "NT step + 20 VALU, %d waves/CU", wpb);
        run(nm2, [&](int it) { hipLaunchKernelGGL((knt<3, 20>), dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl2, 256, 4000);
        snprintf(nm2, sizeof nm2, "NT step + 40 VALU, %d waves/CU", wpb);
        run(nm2, [&](int it) { hipLaunchKernelGGL((knt<3, 40>), dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl2, 256, 4000);
    }
    for (int wpb : {4, 8, 16}) {
        char nm[64];
        auto fl = 36 * 2048.0 * wpb / 4.0;    // run() multiplies by 4 waves per block
        snprintf(nm, sizeof nm, "NT step, reads up front, %d waves/CU", wpb);
        run(nm, [&](int it) { hipLaunchKernelGGL(knt<0>, dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl, 256, 4000);
        snprintf(nm, sizeof nm, "NT step, no LDS reads, %d waves/CU", wpb);
        run(nm, [&](int it) { hipLaunchKernelGGL(knt<1>, dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl, 256, 4000);
        snprintf(nm, sizeof nm, "NT step, production schedule, %d waves/CU", wpb);
        run(nm, [&](int it) { hipLaunchKernelGGL(knt<3>, dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl, 256, 4000);
        snprintf(nm, sizeof nm, "NT step, pair-ahead reads, %d waves/CU", wpb);
        run(nm, [&](int it) { hipLaunchKernelGGL(knt<2>, dim3(256), dim3(64 * wpb), ldsb, 0, out, it, 1.f, w); }, fl, 256, 4000);
    }
    return 0;
}
