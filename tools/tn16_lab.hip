// Lab for the weight-gradient product on the bf16 matrix cores (three-piece split, six piece products):
//   GW[n][k] = sum_m G[m][n] A[m][k]
// Both operands are reduced over their ROW index, so an MFMA lane needs 8 consecutive rows of one
// column: tiles are staged row-major in LDS as bf16 pieces and read back with the transposing
// ds_read_b64_tr_b16.  Build: hipcc -O3 --offload-arch=gfx950 tools/tn16_lab.hip -o tools/bin/tn16_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TR = 32;                 // rows per step
constexpr int PANELS = 18;             // 9 panels of 16 columns for G, 9 for A
constexpr int PANEL_BYTES = 1024 + 32; // 32 row slots x 32 bytes, padded so that the 4 panels a 16-lane group writes differ in banks
constexpr int PIECE_BYTES = PANELS * PANEL_BYTES;
constexpr int BUF_BYTES = 3 * PIECE_BYTES;
constexpr int WAVES = 12, THREADS = 64 * WAVES;

struct Args {
    const float* G; int64_t ldg; int N;
    const float* A; int64_t lda; int K;
    int64_t steps;             // 32-row steps in total
    int sets;                  // workgroups along the rows
    float* partial;            // [sets][81 tiles][4][64]
    float* bias_partial;       // [sets][9 n tiles][4][64] (only lanes with column 0 are meaningful)
    int mode;
};

__device__ __forceinline__ int slot_of(int r) { return (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

__device__ __forceinline__ bf16x8 tr_read(const unsigned char* lds, int addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr + 256));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// One workgroup: all 12 waves stage (3 float4 per thread and step) and multiply.  Wave (wn = id % 3, wk = id / 3) owns
// n tiles 3wn..3wn+2 and k tiles {0,1,2}, {3,4}, {5,6}, {7,8} for wk = 0..3: the SIMDs (waves id, id+4, id+8 share one)
// get 21, 21, 21 and 18 tiles.  Waves with wk == 1 also form the column sums of G.
template <int KT>
__device__ __forceinline__ void tn16_body(const Args& a, unsigned char* lds, const int wn, const int k0t, const bool bias) {
    const int t = threadIdx.x, lane = t & 63;
    const int set = blockIdx.x;
    const int64_t s_lo = a.steps * set / a.sets, s_hi = a.steps * (set + 1) / a.sets;
    const float* src[3]; int64_t ld[3]; int waddr[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int f = t + THREADS * u;
        const int second = f >= 1152, rem = second ? f - 1152 : f;
        const int row = rem / 36, c4 = rem - 36 * row, col = 4 * c4;
        const int width = second ? a.K : a.N;
        const int ceff = col < (width + 3) / 4 * 4 ? col : 0;
        ld[u] = second ? a.lda : a.ldg;
        src[u] = (second ? a.A : a.G) + (int64_t)row * ld[u] + ceff;
        waddr[u] = (second * 9 + (col >> 4)) * PANEL_BYTES + slot_of(row) * 32 + (col & 15) * 2;
    }
    f32x4 stage[3];
    auto load_step = [&](int64_t s) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) stage[u] = *reinterpret_cast<const f32x4*>(src[u] + s * TR * ld[u]);
    };
    auto write_step = [&](int b) __attribute__((always_inline)) {
        unsigned char* base = lds + b * BUF_BYTES;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 hh, mm, ll; split3(stage[u][e], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
            *reinterpret_cast<bf16x4*>(base + waddr[u]) = h;
            *reinterpret_cast<bf16x4*>(base + PIECE_BYTES + waddr[u]) = m;
            *reinterpret_cast<bf16x4*>(base + 2 * PIECE_BYTES + waddr[u]) = l;
        }
    };
    // operand read address inside a panel: lane 4q+p of group kq supplies row 8kq+q, columns 4p..4p+3
    const int kq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int raddr = slot_of(8 * kq + q) * 32 + p * 8;

    f32x4 acc[3][KT], accb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    if (s_lo < s_hi) {
        load_step(s_lo);
        write_step(0);
        if (s_lo + 1 < s_hi) load_step(s_lo + 1);
    }
    __syncthreads();
    for (int64_t s = s_lo; s < s_hi; ++s) {
        const int b = (int)((s - s_lo) & 1);
        const unsigned char* base = lds + b * BUF_BYTES;
        bf16x8 g[3][3], x[KT][3];      // [tile][piece]
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i][pc] = tr_read(base, pc * PIECE_BYTES + (3 * wn + i) * PANEL_BYTES + raddr);
#pragma unroll
            for (int j = 0; j < KT; ++j) x[j][pc] = tr_read(base, pc * PIECE_BYTES + (9 + k0t + j) * PANEL_BYTES + raddr);
        }
        // next step's tiles: split and store while the matrix cores work (other buffer)
        if (s + 1 < s_hi) write_step(b ^ 1);
        if (s + 2 < s_hi) load_step(s + 2);
        // six piece products, small terms first: gl*xh, gh*xl, gm*xm, gh*xm, gm*xh, gh*xh
#define TERM(GP, XP)                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < KT; ++j)                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][GP], x[j][XP], acc[i][j], 0, 0, 0);
        TERM(2, 0) TERM(0, 2) TERM(1, 1) TERM(0, 1) TERM(1, 0) TERM(0, 0)
#undef TERM
        if (bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pc = 2; pc >= 0; --pc) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][pc], ones, accb[i], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: lane (j = lane & 15, q = lane >> 4), register r: n = 16 tn + 4 q + r, k = 16 tk + j
    float* out = a.partial + (int64_t)set * 81 * 256;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(((3 * wn + i) * 9 + k0t + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    if (bias && a.bias_partial) {
        float* bo = a.bias_partial + (int64_t)set * 9 * 256;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) bo[((3 * wn + i) * 4 + r) * 64 + lane] = accb[i][r];
    }
}

__global__ __launch_bounds__(THREADS) void k_tn16(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wk = w / 3, wn = w - 3 * wk;
    if (wk == 0) tn16_body<3>(a, lds, wn, 0, false);
    else tn16_body<2>(a, lds, wn, 1 + 2 * wk, wk == 1);
}

// GW[n][k] = sum over sets of the tile elements + the rows the steps did not cover (fp64)
__global__ void k_reduce(const Args a, int64_t M, float* GW, int64_t ldgw, float* gbias) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t covered = a.steps * TR;
    if (e < (int64_t)a.N * a.K) {
        const int n = (int)(e / a.K), k = (int)(e - (int64_t)n * a.K);
        const int tn = n >> 4, tk = k >> 4, qq = (n & 15) >> 2, r = n & 3, j = k & 15;
        const int64_t off = (((int64_t)tn * 9 + tk) * 4 + r) * 64 + qq * 16 + j;
        double s = 0.0;
        for (int b = 0; b < a.sets; ++b) s += (double)a.partial[(int64_t)b * 81 * 256 + off];
        for (int64_t m = covered; m < M; ++m) s += (double)a.G[m * a.ldg + n] * (double)a.A[m * a.lda + k];
        GW[(int64_t)n * ldgw + k] = (float)s;
    } else if (e < (int64_t)a.N * a.K + a.N && gbias) {
        const int n = (int)(e - (int64_t)a.N * a.K);
        const int tn = n >> 4, qq = (n & 15) >> 2, r = n & 3;
        double s = 0.0;
        for (int b = 0; b < a.sets; ++b) s += (double)a.bias_partial[(int64_t)b * 9 * 256 + ((tn * 4 + r) * 64) + qq * 16];
        for (int64_t m = covered; m < M; ++m) s += (double)a.G[m * a.ldg + n];
        gbias[n] = (float)s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 160984;
    const int N = argc > 2 ? atoi(argv[2]) : 138, K = argc > 3 ? atoi(argv[3]) : 138;
    const int64_t ldg = (N + 3) / 4 * 4, lda = (K + 3) / 4 * 4;
    std::vector<float> hG((size_t)M * ldg), hA((size_t)M * lda);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hG) v = rnd() * 1.7f;
    for (auto& v : hA) v = rnd() * 0.9f + 0.1f;
    float *dG, *dA, *dP, *dB, *dW, *dGb;
    int sets = 256;
    const int64_t steps = (M - 1) / TR;
    if (sets > steps) sets = (int)(steps > 0 ? steps : 1);
    CK(hipMalloc(&dG, hG.size() * 4)); CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dP, (size_t)sets * 81 * 256 * 4)); CK(hipMalloc(&dB, (size_t)sets * 9 * 256 * 4));
    CK(hipMalloc(&dW, (size_t)N * K * 4)); CK(hipMalloc(&dGb, (size_t)N * 4));
    CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    Args a{dG, ldg, N, dA, lda, K, steps, sets, dP, dB, argc > 4 ? atoi(argv[4]) : 0};
    const size_t lds = 2 * BUF_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tn16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_tn16, dim3(sets), dim3(THREADS), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)(((int64_t)N * K + N + 255) / 256)), dim3(256), 0, 0, a, M, dW, (int64_t)K, dGb);
    CK(hipDeviceSynchronize());
    std::vector<float> hW((size_t)N * K), hB(N);
    CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), dGb, hB.size() * 4, hipMemcpyDeviceToHost));
    // reference on a sample of output elements (all of them for small M)
    double maxerr = 0.0, maxref = 0.0, maxberr = 0.0;
    const int stride = M > 20000 ? 97 : 1;
    for (int e = 0; e < N * K; e += stride) {
        const int n = e / K, k = e % K;
        double s = 0.0;
        for (int64_t m = 0; m < M; ++m) s += (double)hG[m * ldg + n] * (double)hA[m * lda + k];
        maxerr = fmax(maxerr, fabs(s - (double)hW[e])); maxref = fmax(maxref, fabs(s));
    }
    for (int n = 0; n < N; n += (stride > 1 ? 7 : 1)) {
        double s = 0.0;
        for (int64_t m = 0; m < M; ++m) s += (double)hG[m * ldg + n];
        maxberr = fmax(maxberr, fabs(s - (double)hB[n]));
    }
    const double fl = 2.0 * M * N * K;
    printf("mode %d M=%lld N=%d K=%d sets=%d: %.1f us  %.1f TF/s-equivalent  max|err|=%.3g (max|ref|=%.3g, rel %.2g)  bias err %.3g\n",
           a.mode, (long long)M, N, K, sets, best * 1e3, fl / best / 1e9, maxerr, maxref, maxerr / maxref, maxberr);
    return 0;
}
