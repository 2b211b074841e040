// fp32 MFMA projections for tall-skinny node-feature matrices (M = nodes, 1e3..1e6; K, N <= ~300).
//
// Replaces the MKL sgemm calls behind SAGEConv.lin_l / lin_r and the decoder Linear layers
// (gvae/gvae_skipcon.py:61-100) and their autograd:
//   forward / grad-input :  C[M,N] = A1[M,K1] W1^T + A2[M,K2] W2^T (+ bias)     (k_gemm_nt)
//                           i.e. the K-concatenated [agg | x] x [W_l | W_r]^T of one SAGE layer,
//                           or [h | skip] x W^T of a decoder level, without materialising the concat;
//                           W is addressed as element(n,k) = W[n*sn + k*sk], so the same kernel
//                           computes g W (grad-input) with (sn, sk) = (1, ld).
//   grad-weight          :  GW[N,K] = G[M,N]^T A[M,K], bias grad = column sums of G  (k_gemm_tn)
//                           reduction over the M rows split over workgroups, partials combined by a
//                           second kernel in a fixed order (no atomics -> reproducible).
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add chain, 64 FLOP/clk/SIMD).
// 16-wide tiles keep padding waste small for the odd widths of this model (138 -> 144, 105 -> 112).
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------
constexpr int G_BM = 128;   // rows per workgroup (32 per wave)
constexpr int G_KC = 32;    // k chunk staged in LDS
constexpr int G_LD = 36;    // LDS row stride in floats (16-byte aligned, spreads ds_read_b128 over banks)

struct GemmSrc {
    const float* A; int64_t lda; int K;
    const float* W; int64_t sn, sk;   // element (n, k) of the weight at W[n*sn + k*sk]
};
struct GemmArgs {
    GemmSrc s[2]; int nsrc;
    const float* bias;
    float* C; int64_t ldc;
    int64_t M; int N;
};

template <int NT>
__global__ __launch_bounds__(256) void k_gemm_nt(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Ws[NT * 16 * G_LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * G_BM;

    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers for the next chunk
    float4 ra[4];
    float rw[2 * NT];
    const int a_c4 = t & 7, a_r = t >> 3;

    // Loads are unconditional on clamped (always valid) addresses; invalid elements are removed by
    // MULTIPLYING with a 0/1 mask when the registers are written to LDS.  (A select "valid ? v : 0"
    // lets the compiler sink each load under its own branch with its own wait, and masking right
    // after the load would force the wait ahead of the MFMAs the load is meant to overlap.)
    // Rows past M reuse row M-1: their results are never stored.
    auto load_chunk = [&](const GemmSrc& s, int kc) {
        const int kb_c = min(kc + 4 * a_c4, ((s.K - 1) >> 2) << 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row_c = min(row0 + a_r + 32 * q, g.M - 1);
            ra[q] = *reinterpret_cast<const float4*>(s.A + row_c * s.lda + kb_c);
        }
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) {
            const int e = t + 256 * q;
            int n, k;
            if (s.sk == 1) { k = e & 31; n = e >> 5; } else { n = e % (NT * 16); k = e / (NT * 16); }
            const int n_c = min(n, g.N - 1), k_c = min(kc + k, s.K - 1);
            rw[q] = s.W[(int64_t)n_c * s.sn + (int64_t)k_c * s.sk];
        }
    };
    auto store_chunk = [&](const GemmSrc& s, int kc) {
        const int kb = kc + 4 * a_c4;
        const float m0 = kb + 0 < s.K ? 1.f : 0.f, m1 = kb + 1 < s.K ? 1.f : 0.f;
        const float m2 = kb + 2 < s.K ? 1.f : 0.f, m3 = kb + 3 < s.K ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = ra[q];
            v.x *= m0; v.y *= m1; v.z *= m2; v.w *= m3;
            *reinterpret_cast<float4*>(&As[(a_r + 32 * q) * G_LD + 4 * a_c4]) = v;
        }
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) {
            const int e = t + 256 * q;
            int n, k;
            if (s.sk == 1) { k = e & 31; n = e >> 5; } else { n = e % (NT * 16); k = e / (NT * 16); }
            const float mk = (n < g.N && kc + k < s.K) ? 1.f : 0.f;
            Ws[n * G_LD + k] = rw[q] * mk;
        }
    };

    auto compute = [&](int kvalid) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (16 * u < kvalid) {
                f32x4 a[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    a[rt] = *reinterpret_cast<const f32x4*>(&As[(32 * w + 16 * rt + i) * G_LD + 16 * u + 4 * kq]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[(16 * nt + i) * G_LD + 16 * u + 4 * kq]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[0][nt] = MFMA16(a[0][m], b[m], acc[0][nt]);
                        acc[1][nt] = MFMA16(a[1][m], b[m], acc[1][nt]);
                    }
                }
            }
        }
    };

    // Software pipeline without loop-carried staging registers: an iteration issues the global loads
    // of chunk c, runs the MFMAs of chunk c-1 out of LDS while they fly, then swaps chunk c into LDS.
    int kvalid_prev = 0;
    for (int si = 0; si < g.nsrc; ++si) {
        const GemmSrc s = g.s[si];
        const int nchunks = (s.K + G_KC - 1) / G_KC;
        for (int c = 0; c < nchunks; ++c) {
            load_chunk(s, c * G_KC);
            if (kvalid_prev > 0) compute(kvalid_prev);
            __syncthreads();
            store_chunk(s, c * G_KC);
            __syncthreads();
            kvalid_prev = min(G_KC, s.K - c * G_KC);
        }
    }
    compute(kvalid_prev);

    // epilogue: D layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + i;
        const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 32 * w + 16 * rt + 4 * kq + r;
                if (row < g.M && col < g.ldc) g.C[row * g.ldc + col] = col < g.N ? acc[rt][nt][r] + bv : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel (weight gradient)
// ------------------------------------------------------------------------------------------
constexpr int T_BM = 32;        // rows (reduction index) per LDS stage

struct WgArgs {
    const float* G; int64_t ldg; int N;
    const float* A; int64_t lda; int K;
    int64_t M; int64_t rows_per_slice;
    float* partial;        // [slices][N*K]
    float* bias_partial;   // [slices][N] or null
};

// NTW: n-tiles per wave (workgroup covers 4*NTW n-tiles), KT: k-tiles per workgroup
template <int NTW, int KT>
__global__ __launch_bounds__(256) void k_gemm_tn(const WgArgs g) {
    constexpr int NB = 4 * NTW * 16;                   // n columns per workgroup
    constexpr int KB = KT * 16;                        // k columns per workgroup
    constexpr int GS_LD = NB + (((NB / 16) & 1) ? 0 : 16);   // == 16 mod 32: rows r, r+1 on disjoint banks
    constexpr int AS_LD = KB + (((KB / 16) & 1) ? 0 : 16);
    __shared__ __attribute__((aligned(16))) float Gs[T_BM * GS_LD];
    __shared__ __attribute__((aligned(16))) float As[T_BM * AS_LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int n_base = blockIdx.z * NB, k_base = blockIdx.y * KB;
    const int64_t m_lo = (int64_t)blockIdx.x * g.rows_per_slice;
    const int64_t m_hi = min(g.M, m_lo + g.rows_per_slice);

    f32x4 acc[NTW][KT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    constexpr int GV = T_BM * (NB / 4);    // float4 pieces of the G stage
    constexpr int AV = T_BM * (KB / 4);
    constexpr int GQ = (GV + 255) / 256, AQ = (AV + 255) / 256;
    float4 rg[GQ], ra[AQ];

    const int n_last = ((g.N - 1) >> 2) << 2, k_last = ((g.K - 1) >> 2) << 2;
    // unconditional loads on clamped addresses; invalid elements are multiplied by a 0 mask when the
    // registers go to LDS (see k_gemm_nt)
    auto load_stage = [&](int64_t m0) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int e = min(t + 256 * q, GV - 1);
            const int r = e / (NB / 4), c4 = e - r * (NB / 4);
            rg[q] = *reinterpret_cast<const float4*>(g.G + min(m0 + r, g.M - 1) * g.ldg + min(n_base + 4 * c4, n_last));
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = min(t + 256 * q, AV - 1);
            const int r = e / (KB / 4), c4 = e - r * (KB / 4);
            ra[q] = *reinterpret_cast<const float4*>(g.A + min(m0 + r, g.M - 1) * g.lda + min(k_base + 4 * c4, k_last));
        }
    };
    auto store_stage = [&](int64_t m0) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int e = t + 256 * q;
            const int r = e / (NB / 4), c4 = e - r * (NB / 4);
            const int col = n_base + 4 * c4;
            const float rm = m0 + r < m_hi ? 1.f : 0.f;
            float4 v = rg[q];
            v.x *= (col + 0 < g.N ? rm : 0.f); v.y *= (col + 1 < g.N ? rm : 0.f);
            v.z *= (col + 2 < g.N ? rm : 0.f); v.w *= (col + 3 < g.N ? rm : 0.f);
            if (e < GV) *reinterpret_cast<float4*>(&Gs[r * GS_LD + 4 * c4]) = v;
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = t + 256 * q;
            const int r = e / (KB / 4), c4 = e - r * (KB / 4);
            const int col = k_base + 4 * c4;
            const float rm = m0 + r < m_hi ? 1.f : 0.f;
            float4 v = ra[q];
            v.x *= (col + 0 < g.K ? rm : 0.f); v.y *= (col + 1 < g.K ? rm : 0.f);
            v.z *= (col + 2 < g.K ? rm : 0.f); v.w *= (col + 3 < g.K ? rm : 0.f);
            if (e < AV) *reinterpret_cast<float4*>(&As[r * AS_LD + 4 * c4]) = v;
        }
    };

    auto compute = [&]() {
        if (g.bias_partial && blockIdx.y == 0 && t < NB) {
#pragma unroll 8
            for (int r = 0; r < T_BM; ++r) bsum += Gs[r * GS_LD + t];
        }
#pragma unroll
        for (int st = 0; st < T_BM / 4; ++st) {
            float ga[NTW], ab[KT];
#pragma unroll
            for (int a = 0; a < NTW; ++a) ga[a] = Gs[(4 * st + kq) * GS_LD + 16 * (w + 4 * a) + i];
#pragma unroll
            for (int b = 0; b < KT; ++b) ab[b] = As[(4 * st + kq) * AS_LD + 16 * b + i];
#pragma unroll
            for (int a = 0; a < NTW; ++a) {
                if (n_base + 16 * (w + 4 * a) < g.N) {   // wave-uniform: tiles past N hold zeros
#pragma unroll
                    for (int b = 0; b < KT; ++b) acc[a][b] = MFMA16(ga[a], ab[b], acc[a][b]);
                }
            }
        }
    };

    // same pipeline shape as k_gemm_nt: load stage s, compute stage s-1, swap
    bool have = false;
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += T_BM) {
        load_stage(m0);
        if (have) compute();
        __syncthreads();
        store_stage(m0);
        __syncthreads();
        have = true;
    }
    if (have) compute();

    float* out = g.partial + (int64_t)blockIdx.x * g.N * g.K;
#pragma unroll
    for (int a = 0; a < NTW; ++a) {
#pragma unroll
        for (int b = 0; b < KT; ++b) {
            const int k = k_base + 16 * b + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n_base + 16 * (w + 4 * a) + 4 * kq + r;
                if (n < g.N && k < g.K) out[(int64_t)n * g.K + k] = acc[a][b][r];
            }
        }
    }
    if (g.bias_partial && blockIdx.y == 0 && t < NB && n_base + t < g.N)
        g.bias_partial[(int64_t)blockIdx.x * g.N + n_base + t] = bsum;
}

// out[e] = sum over slices of partial[s][e].  A workgroup handles 32 elements x 8 slice groups;
// group sums are combined through LDS in a fixed order, so the result is reproducible.
__global__ __launch_bounds__(256) void k_reduce_slices(const float* __restrict__ partial, int slices, int64_t count,
                                                       float* __restrict__ out) {
    __shared__ double sh[8][33];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    double s0 = 0.0, s1 = 0.0;
    if (e < count) {
        int k = grp;
        for (; k + 8 < slices; k += 16) {
            s0 += (double)partial[(int64_t)k * count + e];
            s1 += (double)partial[(int64_t)(k + 8) * count + e];
        }
        if (k < slices) s0 += (double)partial[(int64_t)k * count + e];
    }
    sh[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0 && e < count) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += sh[q][el];
        out[e] = (float)s;
    }
}

template <int NT>
void launch_nt(const GemmArgs& g, hipStream_t st) {
    hipLaunchKernelGGL(k_gemm_nt<NT>, dim3((unsigned)hgv_cdiv(g.M, G_BM)), dim3(256), 0, st, g);
}

bool a_ok(const float* p, int64_t ld, int K) { return p && hgv_aligned(p, 16) && ld % 4 == 0 && ld >= ((K + 3) / 4) * 4; }

template <int NTW, int KT>
void launch_tn(const WgArgs& g, int slices, hipStream_t st) {
    const int gz = (int)hgv_cdiv(g.N, 4 * NTW * 16), gy = (int)hgv_cdiv(g.K, KT * 16);
    hipLaunchKernelGGL((k_gemm_tn<NTW, KT>), dim3(slices, gy, gz), dim3(256), 0, st, g);
}

int tn_slices(int64_t M, int N, int K, int ntw, int kt) {
    const int gz = (int)hgv_cdiv(N, 4 * ntw * 16), gy = (int)hgv_cdiv(K, kt * 16);
    int64_t s = 512 / (gz * gy);
    const int64_t max_s = hgv_cdiv(M, 64);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return (int)s;
}

void pick_tn(int N, int K, int* ntw, int* kt) {
    const int ntiles = (N + 15) / 16, ktiles = (K + 15) / 16;
    *ntw = ntiles <= 4 ? 1 : (ntiles <= 8 ? 2 : 3);
    *kt = ktiles <= 2 ? 2 : (ktiles <= 4 ? 4 : (ktiles <= 7 ? 7 : 9));
}

}  // namespace

// C[M,N] = A1 W1^T (+ A2 W2^T) (+ bias).  Weight element (n,k) of source s is Ws[n*sn_s + k*sk_s]:
// (sn, sk) = (ld, 1) for a row-major [N,K] weight (forward), (1, ld) to multiply by a row-major [K,N]
// matrix (grad-input).  A1/A2 must be 16-byte aligned with lda % 4 == 0 and lda >= ceil(K/4)*4.
// Pad columns [N, ldc) of C are written as zeros.  N <= 288.
extern "C" int hgv_gemm_nt(const float* A1, int64_t lda1, int K1, const float* W1, int64_t sn1, int64_t sk1,
                           const float* A2, int64_t lda2, int K2, const float* W2, int64_t sn2, int64_t sk2,
                           const float* bias, int64_t M, int N, float* C, int64_t ldc, hipStream_t stream) {
    HGV_REQUIRE(M >= 0 && N > 0 && K1 > 0, HGV_EINVAL, "gemm_nt: bad sizes");
    if (M == 0) return HGV_OK;
    HGV_REQUIRE(N <= 288, HGV_EUNSUPPORTED, "gemm_nt: N=%d > 288", N);
    HGV_REQUIRE(a_ok(A1, lda1, K1) && W1 && C && ldc >= N, HGV_EINVAL, "gemm_nt: bad/misaligned buffer (source 1)");
    GemmArgs g{};
    g.s[0] = GemmSrc{A1, lda1, K1, W1, sn1, sk1};
    g.nsrc = 1;
    if (A2) {
        HGV_REQUIRE(K2 > 0 && a_ok(A2, lda2, K2) && W2, HGV_EINVAL, "gemm_nt: bad/misaligned buffer (source 2)");
        g.s[1] = GemmSrc{A2, lda2, K2, W2, sn2, sk2};
        g.nsrc = 2;
    }
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N;
    const int nt = (N + 15) / 16;
    if (nt <= 1) launch_nt<1>(g, stream);
    else if (nt <= 2) launch_nt<2>(g, stream);
    else if (nt <= 4) launch_nt<4>(g, stream);
    else if (nt <= 7) launch_nt<7>(g, stream);
    else if (nt <= 9) launch_nt<9>(g, stream);
    else if (nt <= 12) launch_nt<12>(g, stream);
    else if (nt <= 16) launch_nt<16>(g, stream);
    else launch_nt<18>(g, stream);
    HGV_CHECK_LAUNCH("gemm_nt");
    return HGV_OK;
}

extern "C" size_t hgv_gemm_tn_ws(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    int ntw, kt;
    pick_tn(N, K, &ntw, &kt);
    const int s = tn_slices(M, N, K, ntw, kt);
    return hgv_align256((size_t)s * N * K * 4) + hgv_align256((size_t)s * N * 4);
}

// GW[N,K] (row-major, ld = K) = G[M,N]^T A[M,K];  gbias[N] = column sums of G (nullable).
// G and A must be 16-byte aligned with leading dimensions that are multiples of 4 and cover
// ceil(cols/4)*4 columns.
extern "C" int hgv_gemm_tn(const float* G, int64_t ldg, int N, const float* A, int64_t lda, int K, int64_t M,
                           float* GW, float* gbias, void* ws, size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(M > 0 && N > 0 && K > 0, HGV_EINVAL, "gemm_tn: bad sizes");
    HGV_REQUIRE(a_ok(G, ldg, N) && a_ok(A, lda, K) && GW && ws, HGV_EINVAL, "gemm_tn: bad/misaligned buffer");
    HGV_REQUIRE(ws_bytes >= hgv_gemm_tn_ws(M, N, K) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "gemm_tn: workspace too small or misaligned");
    int ntw, kt;
    pick_tn(N, K, &ntw, &kt);
    const int s = tn_slices(M, N, K, ntw, kt);
    WgArgs g{};
    g.G = G; g.ldg = ldg; g.N = N; g.A = A; g.lda = lda; g.K = K; g.M = M;
    g.rows_per_slice = hgv_cdiv(hgv_cdiv(M, s), T_BM) * T_BM;
    const int slices = (int)hgv_cdiv(M, g.rows_per_slice);
    g.partial = (float*)ws;
    g.bias_partial = gbias ? (float*)((char*)ws + hgv_align256((size_t)s * N * K * 4)) : nullptr;
#define HGV_TN(a, b) if (ntw == a && kt == b) launch_tn<a, b>(g, slices, stream);
    HGV_TN(1, 2) HGV_TN(1, 4) HGV_TN(1, 7) HGV_TN(1, 9)
    HGV_TN(2, 2) HGV_TN(2, 4) HGV_TN(2, 7) HGV_TN(2, 9)
    HGV_TN(3, 2) HGV_TN(3, 4) HGV_TN(3, 7) HGV_TN(3, 9)
#undef HGV_TN
    const int64_t cnt = (int64_t)N * K;
    hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)hgv_cdiv(cnt, 32)), dim3(256), 0, stream, g.partial, slices, cnt, GW);
    if (gbias)
        hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)hgv_cdiv(N, 32)), dim3(256), 0, stream, g.bias_partial, slices,
                           (int64_t)N, gbias);
    HGV_CHECK_LAUNCH("gemm_tn");
    return HGV_OK;
}
