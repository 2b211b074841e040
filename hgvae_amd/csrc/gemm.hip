// fp32 MFMA projections for tall-skinny node-feature matrices (M = nodes, 1e3..1e6; K, N <= ~300).
//
// Replaces the MKL sgemm calls behind SAGEConv.lin_l / lin_r and the decoder Linear layers
// (gvae/gvae_skipcon.py:61-100) and their autograd:
//   forward / grad-input :  C[M,N] = A1[M,K1] W1^T + A2[M,K2] W2^T (+ bias)     (k_gemm_nt)
//                           i.e. the K-concatenated [agg | x] x [W_l | W_r]^T of one SAGE layer,
//                           or [h | skip] x W^T of a decoder level, without materialising the concat.
//                           Weights are first PACKED (k_pack_weight) into a zero-padded, k-contiguous
//                           [ceil16(N)][ceil32(K)] image - from W (forward) or from W^T (grad-input) -
//                           so the GEMM stages them with plain 16-byte loads and needs no masks.
//   grad-weight          :  GW[N,K] = G[M,N]^T A[M,K], bias grad = column sums of G  (k_gemm_tn)
//                           reduction over the M rows split over workgroups, partials combined by a
//                           second kernel in a fixed order (no atomics -> reproducible).
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add chain, 64 FLOP/clk/SIMD).
// 16-wide tiles keep padding waste small for the odd widths of this model (138 -> 144, 105 -> 112).
//
// Staging rule used by both kernels: global loads are unconditional on clamped (always valid)
// addresses and are consumed only when the registers are written to LDS, one pipeline stage later.
// A select "valid ? load : 0" lets the compiler sink each load under its own branch with its own
// wait (serialising them); masking right after the load would put the wait ahead of the MFMAs the
// load is meant to overlap; loop-carried staging registers force a wait at the loop back-edge.
#include "common.h"
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ W, int64_t sn, int64_t sk, int N, int K,
                                                     float* __restrict__ out, int rows_p, int Kp) {
    const int64_t total = (int64_t)rows_p * Kp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / Kp), k = (int)(e - (int64_t)n * Kp);
        out[e] = (n < N && k < K) ? W[(int64_t)n * sn + (int64_t)k * sk] : 0.f;
    }
}

// rows [row_offset, row_offset + N) of a packed image whose other rows the caller keeps zero
__global__ __launch_bounds__(256) void k_pack_weight_rows(const float* __restrict__ W, int64_t sn, int64_t sk, int N,
                                                          int K, float* __restrict__ out, int row_offset, int Kp) {
    const int64_t total = (int64_t)N * Kp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / Kp), k = (int)(e - (int64_t)n * Kp);
        out[(int64_t)(row_offset + n) * Kp + k] = k < K ? W[(int64_t)n * sn + (int64_t)k * sk] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel on packed weights
// ------------------------------------------------------------------------------------------
constexpr int G_BM = 128;   // rows per workgroup (32 per wave)
constexpr int G_KC = 32;    // k chunk staged in LDS
constexpr int G_LD = 36;    // LDS row stride in floats (16-byte aligned, spreads ds_read_b128 over banks)

struct GemmSrc {
    const float* A; int64_t lda; int K;
    const float* Wp; int Kp;           // packed weight [NT*16][Kp], zero outside [N][K]
};
struct GemmArgs {
    GemmSrc s[2]; int nsrc;
    const float* bias;
    float* C; int64_t ldc;
    int64_t M; int N;
};

template <int NT>
__global__ __launch_bounds__(256, (NT <= 9 ? 2 : 1)) void k_gemm_nt(const GemmArgs g) {
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

    constexpr int WV = NT * 16 * 8;              // float4 pieces of one weight chunk
    constexpr int WQ = (WV + 255) / 256;
    f32x4 ra[4], rw[WQ];   // native vectors: arrays of HIP float4 (a struct) captured by the lambdas stay in scratch
    const int c4 = t & 7, r8 = t >> 3;           // this thread's 16-byte column piece / row within 32

    // Rows past M reuse row M-1 (their results are never stored); k pieces past K reuse the last
    // valid piece of A: the packed weight is zero there, so the products vanish.
    auto load_chunk = [&](const GemmSrc s, int kc) __attribute__((always_inline)) {
        const int kb = min(kc + 4 * c4, ((s.K - 1) >> 2) << 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = min(row0 + r8 + 32 * q, g.M - 1);
            ra[q] = *reinterpret_cast<const f32x4*>(s.A + row * s.lda + kb);
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int e = min(t + 256 * q, WV - 1);
            rw[q] = *reinterpret_cast<const f32x4*>(s.Wp + (int64_t)(e >> 3) * s.Kp + kc + 4 * (e & 7));
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&As[(r8 + 32 * q) * G_LD + 4 * c4]) = ra[q];
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int e = t + 256 * q;
            if (e < WV) *reinterpret_cast<f32x4*>(&Ws[(e >> 3) * G_LD + 4 * (e & 7)]) = rw[q];
        }
    };
    auto compute = [&](int kvalid) __attribute__((always_inline)) {
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

    // Software pipeline: an iteration issues the global loads of chunk c, runs the MFMAs of chunk
    // c-1 out of LDS while they fly, then swaps chunk c into LDS.
    int kvalid_prev = 0;
    auto run_source = [&](const GemmSrc s) __attribute__((always_inline)) {
        const int nchunks = (s.K + G_KC - 1) / G_KC;
        for (int c = 0; c < nchunks; ++c) {
            load_chunk(s, c * G_KC);
            if (kvalid_prev > 0) compute(kvalid_prev);
            __syncthreads();
            store_chunk();
            __syncthreads();
            kvalid_prev = min(G_KC, s.K - c * G_KC);
        }
    };
    run_source(g.s[0]);                 // static indices: a runtime-indexed kernel argument would live in scratch
    if (g.nsrc > 1) run_source(g.s[1]);
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

// A workgroup owns NTL x KT output tiles (16x16 each) for one slice of rows; the tiles are dealt
// round-robin to its 4 waves (tile t -> wave t % 4), which balances odd tile counts (9 x 9 = 81).
template <int NTL, int KT>
__global__ __launch_bounds__(256) void k_gemm_tn(const WgArgs g) {
    constexpr int NB = NTL * 16, KB = KT * 16;
    constexpr int GS_LD = NB + ((NTL & 1) ? 0 : 16);   // == 16 mod 32: rows r, r+1 fall on disjoint banks
    constexpr int AS_LD = KB + ((KT & 1) ? 0 : 16);
    constexpr int TT = NTL * KT, TPW = (TT + 3) / 4;
    __shared__ __attribute__((aligned(16))) float Gs[T_BM * GS_LD];
    __shared__ __attribute__((aligned(16))) float As[T_BM * AS_LD];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int n_base = blockIdx.z * NB, k_base = blockIdx.y * KB;
    const int64_t m_lo = (int64_t)blockIdx.x * g.rows_per_slice;
    const int64_t m_hi = min(g.M, m_lo + g.rows_per_slice);

    f32x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    constexpr int GV = T_BM * (NB / 4), AV = T_BM * (KB / 4);   // float4 pieces per stage
    constexpr int GQ = (GV + 255) / 256, AQ = (AV + 255) / 256;
    f32x4 rg[GQ], ra[AQ];
    const int n_last = ((g.N - 1) >> 2) << 2, k_last = ((g.K - 1) >> 2) << 2;

    auto load_stage = [&](int64_t m0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int e = min(t + 256 * q, GV - 1);
            const int r = e / (NB / 4), c4 = e - r * (NB / 4);
            rg[q] = *reinterpret_cast<const f32x4*>(g.G + min(m0 + r, g.M - 1) * g.ldg + min(n_base + 4 * c4, n_last));
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = min(t + 256 * q, AV - 1);
            const int r = e / (KB / 4), c4 = e - r * (KB / 4);
            ra[q] = *reinterpret_cast<const f32x4*>(g.A + min(m0 + r, g.M - 1) * g.lda + min(k_base + 4 * c4, k_last));
        }
    };
    // invalid rows / columns are zeroed by multiplying with a 0/1 mask on the way to LDS
    auto store_stage = [&](int64_t m0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int e = t + 256 * q;
            const int r = e / (NB / 4), c4 = e - r * (NB / 4);
            const int col = n_base + 4 * c4;
            const float rm = m0 + r < m_hi ? 1.f : 0.f;
            const f32x4 mk = {col + 0 < g.N ? rm : 0.f, col + 1 < g.N ? rm : 0.f, col + 2 < g.N ? rm : 0.f,
                              col + 3 < g.N ? rm : 0.f};
            if (e < GV) *reinterpret_cast<f32x4*>(&Gs[r * GS_LD + 4 * c4]) = rg[q] * mk;
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = t + 256 * q;
            const int r = e / (KB / 4), c4 = e - r * (KB / 4);
            const int col = k_base + 4 * c4;
            const float rm = m0 + r < m_hi ? 1.f : 0.f;
            const f32x4 mk = {col + 0 < g.K ? rm : 0.f, col + 1 < g.K ? rm : 0.f, col + 2 < g.K ? rm : 0.f,
                              col + 3 < g.K ? rm : 0.f};
            if (e < AV) *reinterpret_cast<f32x4*>(&As[r * AS_LD + 4 * c4]) = ra[q] * mk;
        }
    };

    // Wave-specialised MFMA body: with W a compile-time constant every tile's (n-tile, k-tile) pair
    // is static, so the accumulators stay in registers and unused fragment loads disappear.
    auto compute_w = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
#pragma unroll
        for (int st = 0; st < T_BM / 4; ++st) {
            float ga[NTL], ab[KT];
#pragma unroll
            for (int a = 0; a < NTL; ++a) ga[a] = Gs[(4 * st + kq) * GS_LD + 16 * a + i];
#pragma unroll
            for (int b = 0; b < KT; ++b) ab[b] = As[(4 * st + kq) * AS_LD + 16 * b + i];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int tile = W + 4 * j;
                if (tile < TT) acc[j] = MFMA16(ga[tile / KT], ab[tile % KT], acc[j]);
            }
        }
    };
    auto compute = [&]() __attribute__((always_inline)) {
        if (g.bias_partial && blockIdx.y == 0 && t < NB) {
#pragma unroll 8
            for (int r = 0; r < T_BM; ++r) bsum += Gs[r * GS_LD + t];
        }
        if (w == 0) compute_w(std::integral_constant<int, 0>{});
        else if (w == 1) compute_w(std::integral_constant<int, 1>{});
        else if (w == 2) compute_w(std::integral_constant<int, 2>{});
        else compute_w(std::integral_constant<int, 3>{});
    };

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
    for (int j = 0; j < TPW; ++j) {
        const int tile = w + 4 * j;
        if (tile < TT) {
            const int a = tile / KT, b = tile - a * KT;
            const int k = k_base + 16 * b + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n_base + 16 * a + 4 * kq + r;
                if (n < g.N && k < g.K) out[(int64_t)n * g.K + k] = acc[j][r];
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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int nt_template(int N) {   // instantiated NT >= ceil(N / 16)
    const int nt = (N + 15) / 16;
    return nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : nt <= 7 ? 7 : nt <= 9 ? 9 : nt <= 12 ? 12 : nt <= 16 ? 16 : 18;
}
int kp_of(int K) { return (K + 31) / 32 * 32; }

template <int NT>
void launch_nt(const GemmArgs& g, hipStream_t st) {
    hipLaunchKernelGGL(k_gemm_nt<NT>, dim3((unsigned)hgv_cdiv(g.M, G_BM)), dim3(256), 0, st, g);
}

bool a_ok(const float* p, int64_t ld, int K) { return p && hgv_aligned(p, 16) && ld % 4 == 0 && ld >= ((K + 3) / 4) * 4; }

void pick_tn(int N, int K, int* ntl, int* kt) {
    const int ntiles = (N + 15) / 16, ktiles = (K + 15) / 16;
    *ntl = ntiles <= 1 ? 1 : ntiles <= 2 ? 2 : ntiles <= 4 ? 4 : ntiles <= 7 ? 7 : ntiles <= 9 ? 9 : 12;
    *kt = ktiles <= 2 ? 2 : (ktiles <= 4 ? 4 : (ktiles <= 7 ? 7 : 9));
}

int tn_slices(int64_t M, int N, int K, int ntl, int kt) {
    const int gz = (int)hgv_cdiv(N, ntl * 16), gy = (int)hgv_cdiv(K, kt * 16);
    int64_t s = 512 / (gz * gy);
    const int64_t max_s = hgv_cdiv(M, 64);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return (int)s;
}

template <int NTL, int KT>
void launch_tn(const WgArgs& g, int slices, hipStream_t st) {
    const int gz = (int)hgv_cdiv(g.N, NTL * 16), gy = (int)hgv_cdiv(g.K, KT * 16);
    hipLaunchKernelGGL((k_gemm_tn<NTL, KT>), dim3(slices, gy, gz), dim3(256), 0, st, g);
}

}  // namespace

// Packed image of a weight for hgv_gemm_nt_packed: [rows][Kp] floats, rows = hgv_pack_rows(N),
// Kp = ceil(K/32)*32, element (n,k) = W[n*sn + k*sk] inside [N][K] and zero outside.
extern "C" int hgv_pack_rows(int N) { return nt_template(N) * 16; }
extern "C" size_t hgv_pack_bytes(int N, int K) { return hgv_align256((size_t)hgv_pack_rows(N) * kp_of(K) * 4); }

extern "C" int hgv_pack_weight(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, hipStream_t stream) {
    HGV_REQUIRE(W && out && N > 0 && K > 0, HGV_EINVAL, "pack_weight: bad arguments");
    HGV_REQUIRE(N <= 288, HGV_EUNSUPPORTED, "pack_weight: N=%d > 288", N);
    HGV_REQUIRE(hgv_aligned(out, 16), HGV_EINVAL, "pack_weight: output must be 16-byte aligned");
    const int rows = hgv_pack_rows(N), Kp = kp_of(K);
    const int64_t total = (int64_t)rows * Kp;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)hgv_cdiv(total, 256)), dim3(256), 0, stream, W, sn, sk, N, K, out, rows, Kp);
    HGV_CHECK_LAUNCH("pack_weight");
    return HGV_OK;
}

// Writes one weight into rows [row_offset, row_offset + N) of a packed image that holds several
// weights stacked along N (e.g. [W_l ; W_r] so that one product yields both halves).  The image has
// hgv_pack_rows(total N) rows of ceil(K/32)*32 floats; rows not covered by any weight must be zero
// (zero the buffer once when it is allocated).
extern "C" int hgv_pack_weight_rows(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, int row_offset,
                                    hipStream_t stream) {
    HGV_REQUIRE(W && out && N > 0 && K > 0 && row_offset >= 0, HGV_EINVAL, "pack_weight_rows: bad arguments");
    HGV_REQUIRE(row_offset + N <= 288, HGV_EUNSUPPORTED, "pack_weight_rows: more than 288 rows");
    const int Kp = kp_of(K);
    hipLaunchKernelGGL(k_pack_weight_rows, dim3((unsigned)hgv_cdiv((int64_t)N * Kp, 256)), dim3(256), 0, stream, W, sn, sk,
                       N, K, out, row_offset, Kp);
    HGV_CHECK_LAUNCH("pack_weight_rows");
    return HGV_OK;
}

// C[M,N] = A1 Wp1^T (+ A2 Wp2^T) (+ bias) with packed weights (hgv_pack_weight).
// A1/A2: 16-byte aligned, lda % 4 == 0, lda >= ceil(K/4)*4, pad columns [K, lda) finite (zero in this
// library's layout).  Columns [N, ldc) of C are written as zeros.  N <= 288.
extern "C" int hgv_gemm_nt_packed(const float* A1, int64_t lda1, int K1, const float* Wp1, const float* A2,
                                  int64_t lda2, int K2, const float* Wp2, const float* bias, int64_t M, int N,
                                  float* C, int64_t ldc, hipStream_t stream) {
    HGV_REQUIRE(M >= 0 && N > 0 && K1 > 0, HGV_EINVAL, "gemm_nt: bad sizes");
    if (M == 0) return HGV_OK;
    HGV_REQUIRE(N <= 288, HGV_EUNSUPPORTED, "gemm_nt: N=%d > 288", N);
    HGV_REQUIRE(a_ok(A1, lda1, K1) && Wp1 && hgv_aligned(Wp1, 16) && C && ldc >= N, HGV_EINVAL,
                "gemm_nt: bad/misaligned buffer (source 1)");
    GemmArgs g{};
    g.s[0] = GemmSrc{A1, lda1, K1, Wp1, kp_of(K1)};
    g.nsrc = 1;
    if (A2) {
        HGV_REQUIRE(K2 > 0 && a_ok(A2, lda2, K2) && Wp2 && hgv_aligned(Wp2, 16), HGV_EINVAL,
                    "gemm_nt: bad/misaligned buffer (source 2)");
        g.s[1] = GemmSrc{A2, lda2, K2, Wp2, kp_of(K2)};
        g.nsrc = 2;
    }
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N;
    switch (nt_template(N)) {
        case 1: launch_nt<1>(g, stream); break;
        case 2: launch_nt<2>(g, stream); break;
        case 4: launch_nt<4>(g, stream); break;
        case 7: launch_nt<7>(g, stream); break;
        case 9: launch_nt<9>(g, stream); break;
        case 12: launch_nt<12>(g, stream); break;
        case 16: launch_nt<16>(g, stream); break;
        default: launch_nt<18>(g, stream); break;
    }
    HGV_CHECK_LAUNCH("gemm_nt");
    return HGV_OK;
}

// Convenience form: packs the weights into ws, then multiplies.  Weight element (n,k) of source s is
// Ws[n*sn_s + k*sk_s]: (sn, sk) = (ld, 1) for a row-major [N,K] weight (forward), (1, ld) to multiply by
// a row-major [K,N] matrix (grad-input).
extern "C" size_t hgv_gemm_nt_ws(int N, int K1, int K2) {
    return hgv_pack_bytes(N, K1) + (K2 > 0 ? hgv_pack_bytes(N, K2) : 0);
}

extern "C" int hgv_gemm_nt(const float* A1, int64_t lda1, int K1, const float* W1, int64_t sn1, int64_t sk1,
                           const float* A2, int64_t lda2, int K2, const float* W2, int64_t sn2, int64_t sk2,
                           const float* bias, int64_t M, int N, float* C, int64_t ldc, void* ws, size_t ws_bytes,
                           hipStream_t stream) {
    HGV_REQUIRE(N > 0 && K1 > 0 && W1 && ws, HGV_EINVAL, "gemm_nt: bad arguments");
    HGV_REQUIRE(ws_bytes >= hgv_gemm_nt_ws(N, K1, A2 ? K2 : 0) && hgv_aligned(ws, 256), HGV_EWORKSPACE,
                "gemm_nt: workspace too small or misaligned");
    if (M == 0) return HGV_OK;
    float* p1 = (float*)ws;
    float* p2 = (float*)((char*)ws + hgv_pack_bytes(N, K1));
    int rc = hgv_pack_weight(W1, sn1, sk1, N, K1, p1, stream);
    if (rc != HGV_OK) return rc;
    if (A2) {
        HGV_REQUIRE(W2 && K2 > 0, HGV_EINVAL, "gemm_nt: source 2 incomplete");
        rc = hgv_pack_weight(W2, sn2, sk2, N, K2, p2, stream);
        if (rc != HGV_OK) return rc;
    }
    return hgv_gemm_nt_packed(A1, lda1, K1, p1, A2, lda2, K2, A2 ? p2 : nullptr, bias, M, N, C, ldc, stream);
}

extern "C" size_t hgv_gemm_tn_ws(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    int ntl, kt;
    pick_tn(N, K, &ntl, &kt);
    const int s = tn_slices(M, N, K, ntl, kt);
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
    int ntl, kt;
    pick_tn(N, K, &ntl, &kt);
    const int s = tn_slices(M, N, K, ntl, kt);
    WgArgs g{};
    g.G = G; g.ldg = ldg; g.N = N; g.A = A; g.lda = lda; g.K = K; g.M = M;
    g.rows_per_slice = hgv_cdiv(hgv_cdiv(M, s), T_BM) * T_BM;
    const int slices = (int)hgv_cdiv(M, g.rows_per_slice);
    g.partial = (float*)ws;
    g.bias_partial = gbias ? (float*)((char*)ws + hgv_align256((size_t)s * N * K * 4)) : nullptr;
#define HGV_TN(a, b) if (ntl == a && kt == b) launch_tn<a, b>(g, slices, stream);
    HGV_TN(1, 2) HGV_TN(1, 4) HGV_TN(1, 7) HGV_TN(1, 9)
    HGV_TN(2, 2) HGV_TN(2, 4) HGV_TN(2, 7) HGV_TN(2, 9)
    HGV_TN(4, 2) HGV_TN(4, 4) HGV_TN(4, 7) HGV_TN(4, 9)
    HGV_TN(7, 2) HGV_TN(7, 4) HGV_TN(7, 7) HGV_TN(7, 9)
    HGV_TN(9, 2) HGV_TN(9, 4) HGV_TN(9, 7) HGV_TN(9, 9)
    HGV_TN(12, 2) HGV_TN(12, 4) HGV_TN(12, 7) HGV_TN(12, 9)
#undef HGV_TN
    const int64_t cnt = (int64_t)N * K;
    hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)hgv_cdiv(cnt, 32)), dim3(256), 0, stream, g.partial, slices, cnt, GW);
    if (gbias)
        hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)hgv_cdiv(N, 32)), dim3(256), 0, stream, g.bias_partial, slices,
                           (int64_t)N, gbias);
    HGV_CHECK_LAUNCH("gemm_tn");
    return HGV_OK;
}
