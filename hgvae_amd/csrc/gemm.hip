// fp32 MFMA projections for tall-skinny node-feature matrices (M = nodes, 1e3..1e6; K, N <= ~300).
//
// Replaces the MKL sgemm calls behind SAGEConv.lin_l / lin_r and the decoder Linear layers
// (gvae/gvae_skipcon.py:61-100) and their autograd:
//   forward / grad-input :  C[M,N] = A1[M,K1] W1^T + A2[M,K2] W2^T (+ bias)     (k_gemm_nt)
//                           i.e. the K-concatenated [agg | x] x [W_l | W_r]^T of one SAGE layer,
//                           or [h | skip] x W^T of a decoder level, without materialising the concat.
//                           Weights are first PACKED (k_pack_weight) into a zero-padded image ordered as the
//                           MFMA consumes it - from W (forward) or from W^T (grad-input).  When the image
//                           fits in LDS (<= 160 KB, true for every layer of the 138-wide model) it stays
//                           resident there and waves stream rows of A without barriers (k_gemm_nt_res);
//                           larger weights are staged per 32-wide k chunk (k_gemm_nt).
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
// Packed image of a weight: [Gp][R][4] floats, where group g holds k = 4g..4g+3, R = 16 * (n tiles)
// rows and Gp = ceil(K/32)*8 groups; element (n,k) = W[n*sn + k*sk] inside [N][K] and zero outside.
// An MFMA lane (i, kq) reads the 16 bytes of (group, row i) as one operand quad, so a wave's
// ds_read_b128 touches 16 consecutive 16-byte slots per k group: conflict-free without padding.
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ W, int64_t sn, int64_t sk, int N, int K,
                                                     float* __restrict__ out, int R, int Gp) {
    const int64_t total = (int64_t)Gp * R * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int gidx = (int)(e / (R * 4)), rem = (int)(e - (int64_t)gidx * R * 4);
        const int n = rem >> 2, k = 4 * gidx + (rem & 3);
        out[e] = (n < N && k < K) ? W[(int64_t)n * sn + (int64_t)k * sk] : 0.f;
    }
}

// rows [row_offset, row_offset + N) of a packed image with R rows whose other rows the caller keeps zero
__global__ __launch_bounds__(256) void k_pack_weight_rows(const float* __restrict__ W, int64_t sn, int64_t sk, int N,
                                                          int K, float* __restrict__ out, int row_offset, int R, int Gp) {
    const int64_t total = (int64_t)Gp * N * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int gidx = (int)(e / (N * 4)), rem = (int)(e - (int64_t)gidx * N * 4);
        const int n = rem >> 2, k = 4 * gidx + (rem & 3);
        out[((int64_t)gidx * R + row_offset + n) * 4 + (rem & 3)] = k < K ? W[(int64_t)n * sn + (int64_t)k * sk] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel, weights resident in LDS
// ------------------------------------------------------------------------------------------
// The whole packed weight (both sources) is copied into LDS once per workgroup; after that single
// barrier every wave streams 16-row tiles of A on its own: A operand quads come straight from
// global memory into registers (a ring of RING loads in flight per wave, running ahead across tile
// boundaries), weight quads from LDS.  No barriers and no LDS traffic for A in the main loop.
// The MFMA operands are swapped (weights as the row operand), so a lane ends up with 4 consecutive
// output columns of one row and stores them as one 16-byte piece.
struct ResSrc {
    const float* A; int64_t lda; int G;    // G = ceil(K/4) k groups
    const float* Wp;                       // packed image (R = NT*16 rows)
};
struct ResArgs {
    ResSrc s[2]; int nsrc;
    const float* bias;
    float* C; int64_t ldc;
    int64_t M; int N; int64_t ntiles;
};

__device__ __forceinline__ int res_quads(int G) {   // MFMA quads (4 k groups each) needed for G groups
    const int rem = G & 7;
    return 2 * (G >> 3) + (rem > 4 ? 2 : (rem > 0 ? 1 : 0));
}
// k group handled by lanes kq of quad q: full blocks of 8 groups are split so that a lane's two
// consecutive quads read adjacent 16-byte pieces (a wave then covers 128 contiguous bytes per row);
// a tail of <= 4 groups is one plain quad.
__device__ __forceinline__ int res_group(int G, int q, int kq) {
    const int P = G >> 3, rem = G & 7;
    return (q < 2 * P || rem > 4) ? 8 * (q >> 1) + 2 * kq + (q & 1) : 8 * P + kq;
}

// Instruction order of one quad step, imposed on the scheduler: the LDS reads of the first two
// pairs of n tiles, then per pair its MFMAs followed by the reads of the pair two ahead, so that
// no MFMA waits for a read issued right before it.
template <int NT, int RT, int NTI>
struct ResSched {
    static __device__ __forceinline__ void run() {
        if constexpr (NTI < NT) {
            __builtin_amdgcn_sched_group_barrier(0x008, (NTI + 1 < NT ? 8 : 4) * RT, 0);
            if constexpr (NTI + 4 < NT) __builtin_amdgcn_sched_group_barrier(0x100, NTI + 5 < NT ? 2 : 1, 0);
            ResSched<NT, RT, NTI + 2>::run();
        }
    }
};

template <int NT, int RT, int RING>
__global__ __launch_bounds__(1024) void k_gemm_nt_res(const ResArgs g) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];   // [G1 + G2][R][4], bias[R], step table[Qp][4]
    constexpr int R = NT * 16;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6), wpb = blockDim.x >> 6;
    const float* A1 = g.s[0].A; const int64_t lda1 = g.s[0].lda; const int G1 = g.s[0].G;
    const float* A2 = g.nsrc > 1 ? g.s[1].A : g.s[0].A;
    const int64_t lda2 = g.nsrc > 1 ? g.s[1].lda : g.s[0].lda;
    const int G2 = g.nsrc > 1 ? g.s[1].G : 0;
    const int nq1 = res_quads(G1), Q = nq1 + res_quads(G2);
    const int Qp = (Q + RING - 1) / RING * RING;     // steps per tile, padded to whole ring rounds
    float* bias_s = Wl + (int64_t)(G1 + G2) * R * 4;
    int2* tab = reinterpret_cast<int2*>(bias_s + R);
    {
        const int n1 = G1 * R, n2 = G2 * R;     // 16-byte pieces
        f32x4* dst = reinterpret_cast<f32x4*>(Wl);
        const f32x4* w1 = reinterpret_cast<const f32x4*>(g.s[0].Wp);
        for (int e = t; e < n1; e += blockDim.x) dst[e] = w1[e];
        if (n2 > 0) {
            const f32x4* w2 = reinterpret_cast<const f32x4*>(g.s[1].Wp);
            for (int e = t; e < n2; e += blockDim.x) dst[n1 + e] = w2[e];
        }
        for (int e = t; e < R; e += blockDim.x) bias_s[e] = (g.bias && e < g.N) ? g.bias[e] : 0.f;
        // Step table: for step fq of a tile and lane group kq, .x = byte offset of the A quad in its
        // row (bit 31 set: the quad is padding and counts as zero), .y = byte offset of the weight
        // group in LDS.  Padding quads point at valid memory (the last group) so loads stay in bounds.
        for (int e = t; e < Qp * 4; e += blockDim.x) {
            const int fq = min(e >> 2, Q - 1), k = e & 3;
            const bool second = fq >= nq1;
            const int G = second ? G2 : G1;
            const int gi = res_group(G, second ? fq - nq1 : fq, k);
            const int gc = min(gi, G - 1);
            const bool keep = gi < G && (e >> 2) < Q;
            tab[e] = make_int2((gc * 16) | (keep ? 0 : (int)0x80000000), ((second ? G1 : 0) + gc) * R * 16);
        }
    }
    __syncthreads();

    // tiles are dealt so that the waves holding one tile more than the others spread over all workgroups
    const int64_t gw = (int64_t)wib * gridDim.x + blockIdx.x, nw = (int64_t)wpb * gridDim.x;
    const int64_t my_tiles = gw < g.ntiles ? (g.ntiles - gw + nw - 1) / nw : 0;

    f32x4 acc[RT][NT];
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(&bias_s[16 * nt + 4 * kq]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = b;
        }
    };
    auto compute = [&](const f32x4 (&a_in)[RT], const int2 e) __attribute__((always_inline)) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wl) + e.y) + i;
        // weight quads are read one pair of n tiles ahead of the MFMAs that use them
        f32x4 b0 = wp[0], b1 = wp[NT > 1 ? 16 : 0];
        f32x4 a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = e.x >= 0 ? a_in[rt] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
            f32x4 n0 = b0, n1 = b1;
            if (nt + 2 < NT) n0 = wp[16 * (nt + 2)];
            if (nt + 3 < NT) n1 = wp[16 * (nt + 3)];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][nt] = MFMA16(b0[m], a[rt][m], acc[rt][nt]);
                    if (nt + 1 < NT) acc[rt][nt + 1] = MFMA16(b1[m], a[rt][m], acc[rt][nt + 1]);
                }
            }
            b0 = n0; b1 = n1;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NT < 4 ? NT : 4, 0);
        ResSched<NT, RT, 0>::run();
    };
    // D layout with the weights as the row operand: lane (i, kq) register r = C[row i][16 nt + 4 kq + r]
    auto store_tile = [&](int64_t tile) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = tile * (16 * RT) + 16 * rt + i;
            if (row < g.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = 16 * nt + 4 * kq;
                    if (col < g.ldc) *reinterpret_cast<f32x4*>(g.C + row * g.ldc + col) = acc[rt][nt];
                }
            }
        }
    };

    // Load cursor: runs RING steps ahead of the MFMAs, across tile boundaries.  Every ring slot
    // issues exactly one load group per step (past the end: a repeat of the last one), so the number
    // of loads in flight is the same on every path and each wait is a counted one.
    // Rows past M reuse row M-1 (never stored).
    int64_t ltile = gw, lidx = 0; int lq = 0;
    const char* lp1[RT]; const char* lp2[RT];
    auto row_pointers = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = min(ltile * (16 * RT) + 16 * rt + i, g.M - 1);
            lp1[rt] = reinterpret_cast<const char*>(A1 + row * lda1);
            lp2[rt] = reinterpret_cast<const char*>(A2 + row * lda2);
        }
    };
    f32x4 ring[RING][RT];
    int2 ring_e[RING];
    int2 e_next = make_int2(0, 0);
    auto issue = [&](int d) __attribute__((always_inline)) {
        const int2 e = e_next;
        const bool second = min(lq, Q - 1) >= nq1;
        ring_e[d] = e;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            ring[d][rt] = *reinterpret_cast<const f32x4*>((second ? lp2[rt] : lp1[rt]) + (e.x & 0x7fffffff));
        if (++lq == Qp) {
            if (lidx + 1 < my_tiles) { lq = 0; ++lidx; ltile += nw; row_pointers(); } else lq = Qp - 1;
        }
        e_next = tab[lq * 4 + kq];
    };
    if (my_tiles > 0) {
        row_pointers();
        e_next = tab[kq];
#pragma unroll
        for (int d = 0; d < RING; ++d) issue(d);
    }
    int64_t ctile = gw;
    for (int64_t ti = 0; ti < my_tiles; ++ti, ctile += nw) {
        init_acc();
        for (int q0 = 0; q0 < Qp; q0 += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = ring[d][rt];
                const int2 e = ring_e[d];
                issue(d);
                compute(a, e);
            }
        }
        store_tile(ctile);
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel, weights staged per 32-wide k chunk (for weights too large to keep in LDS)
// ------------------------------------------------------------------------------------------
constexpr int G_KC = 32;    // k chunk staged in LDS
constexpr int G_LD = 36;    // LDS row stride of the A tile in floats (16-byte aligned, spreads ds_read_b128 over banks)

struct GemmSrc {
    const float* A; int64_t lda; int K;
    const float* Wp;                   // packed weight image, R = NT*16 rows
};
struct GemmArgs {
    GemmSrc s[2]; int nsrc;
    const float* bias;
    float* C; int64_t ldc;
    int64_t M; int N;
};

// RT: 16-row tiles per wave (workgroup = 4 waves = 64*RT rows); MINW: occupancy hint (waves per SIMD)
template <int NT, int RT, int MINW>
__global__ __launch_bounds__(256, MINW) void k_gemm_nt(const GemmArgs g) {
    constexpr int G_BM = 64 * RT;
    constexpr int R = NT * 16;
    __shared__ __attribute__((aligned(16))) float As[G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Ws[R * G_KC];      // [8 groups][R][4]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * G_BM;

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int WV = R * 8;                    // 16-byte pieces of one weight chunk (contiguous in the image)
    constexpr int WQ = (WV + 255) / 256;
    constexpr int AQ = G_BM / 32;
    f32x4 ra[AQ], rw[WQ];   // native vectors: arrays of HIP float4 (a struct) captured by the lambdas stay in scratch
    const int c4 = t & 7, r8 = t >> 3;           // this thread's 16-byte column piece / row within 32

    // Rows past M reuse row M-1 (their results are never stored); k pieces past K reuse the last
    // valid piece of A: the packed weight is zero there, so the products vanish.
    auto load_chunk = [&](const GemmSrc s, int kc) __attribute__((always_inline)) {
        const int kb = min(kc + 4 * c4, ((s.K - 1) >> 2) << 2);
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int64_t row = min(row0 + r8 + 32 * q, g.M - 1);
            ra[q] = *reinterpret_cast<const f32x4*>(s.A + row * s.lda + kb);
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(s.Wp) + (int64_t)(kc >> 2) * R;
#pragma unroll
        for (int q = 0; q < WQ; ++q) rw[q] = wsrc[min(t + 256 * q, WV - 1)];
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) *reinterpret_cast<f32x4*>(&As[(r8 + 32 * q) * G_LD + 4 * c4]) = ra[q];
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int e = t + 256 * q;
            if (e < WV) *reinterpret_cast<f32x4*>(&Ws[4 * e]) = rw[q];
        }
    };
    auto compute = [&](int kvalid) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (16 * u < kvalid) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a[rt] = *reinterpret_cast<const f32x4*>(&As[(16 * RT * w + 16 * rt + i) * G_LD + 16 * u + 4 * kq]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[((4 * u + kq) * R + 16 * nt + i) * 4]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = MFMA16(a[rt][m], b[m], acc[rt][nt]);
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
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 16 * RT * w + 16 * rt + 4 * kq + r;
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
    constexpr int W2 = NT <= 9 ? 2 : 1;
    hipLaunchKernelGGL((k_gemm_nt<NT, 1, W2>), dim3((unsigned)hgv_cdiv(g.M, 64)), dim3(256), 0, st, g);
}

constexpr size_t RES_LDS_MAX = 160 * 1024;
int host_quads(int G) { const int rem = G & 7; return 2 * (G >> 3) + (rem > 4 ? 2 : (rem > 0 ? 1 : 0)); }
size_t res_lds_bytes(int R, int G1, int G2) {   // weights, bias, step table (<= 3 padding steps)
    return ((size_t)(G1 + G2) * R * 4 + R) * sizeof(float) + (size_t)(host_quads(G1) + host_quads(G2) + 3) * 4 * 8;
}

int cu_count() {   // compute units of the current device (cached per device)
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

template <int NT, int RING>
int launch_nt_res(const ResArgs& g, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt_res<NT, 1, RING>;
    static bool raised[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised[dev]) {   // dynamic LDS above the default limit has to be requested once per device
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS_MAX);
        HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "gemm_nt: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        raised[dev] = true;
    }
    // One workgroup per compute unit; as many waves (<= 16) as there are tiles to hand out.
    const int cus = cu_count();
    int64_t wpb = hgv_cdiv(g.ntiles, cus);
    if (wpb > 16) wpb = 16;
    if (wpb < 1) wpb = 1;
    int64_t blocks = hgv_cdiv(g.ntiles, wpb);
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((unsigned)(64 * wpb)), lds, st, g);
    return HGV_OK;
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

// Packed image of a weight for hgv_gemm_nt_packed: [Gp][rows][4] floats with rows = hgv_pack_rows(N),
// Gp = ceil(K/32)*8 groups of 4 consecutive k; element (n,k) = W[n*sn + k*sk] inside [N][K], zero outside.
extern "C" int hgv_pack_rows(int N) { return nt_template(N) * 16; }
extern "C" size_t hgv_pack_bytes(int N, int K) { return hgv_align256((size_t)hgv_pack_rows(N) * kp_of(K) * 4); }

extern "C" int hgv_pack_weight(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, hipStream_t stream) {
    HGV_REQUIRE(W && out && N > 0 && K > 0, HGV_EINVAL, "pack_weight: bad arguments");
    HGV_REQUIRE(N <= 288, HGV_EUNSUPPORTED, "pack_weight: N=%d > 288", N);
    HGV_REQUIRE(hgv_aligned(out, 16), HGV_EINVAL, "pack_weight: output must be 16-byte aligned");
    const int rows = hgv_pack_rows(N), Gp = kp_of(K) / 4;
    const int64_t total = (int64_t)rows * Gp * 4;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)hgv_cdiv(total, 256)), dim3(256), 0, stream, W, sn, sk, N, K, out, rows, Gp);
    HGV_CHECK_LAUNCH("pack_weight");
    return HGV_OK;
}

// Writes one weight into rows [row_offset, row_offset + N) of a packed image that holds several
// weights stacked along N (e.g. [W_l ; W_r] so that one product yields both halves).  The image is
// the one of a weight with n_total rows (hgv_pack_bytes(n_total, K) bytes); rows not covered by any
// weight must be zero (zero the buffer once when it is allocated).
extern "C" int hgv_pack_weight_rows(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, int row_offset,
                                    int n_total, hipStream_t stream) {
    HGV_REQUIRE(W && out && N > 0 && K > 0 && row_offset >= 0, HGV_EINVAL, "pack_weight_rows: bad arguments");
    HGV_REQUIRE(row_offset + N <= n_total && n_total <= 288, HGV_EUNSUPPORTED,
                "pack_weight_rows: rows [%d, %d) outside an image of %d (<= 288) rows", row_offset, row_offset + N, n_total);
    const int Gp = kp_of(K) / 4;
    hipLaunchKernelGGL(k_pack_weight_rows, dim3((unsigned)hgv_cdiv((int64_t)N * Gp * 4, 256)), dim3(256), 0, stream, W, sn,
                       sk, N, K, out, row_offset, hgv_pack_rows(n_total), Gp);
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
    if (A2)
        HGV_REQUIRE(K2 > 0 && a_ok(A2, lda2, K2) && Wp2 && hgv_aligned(Wp2, 16), HGV_EINVAL,
                    "gemm_nt: bad/misaligned buffer (source 2)");
    const int nt = nt_template(N), R = nt * 16;
    const int G1 = (K1 + 3) / 4, G2 = A2 ? (K2 + 3) / 4 : 0;
    const size_t lds = res_lds_bytes(R, G1, G2);
    if (lds <= RES_LDS_MAX && hgv_aligned(C, 16) && ldc % 4 == 0) {
        ResArgs g{};
        g.s[0] = ResSrc{A1, lda1, G1, Wp1};
        g.nsrc = 1;
        if (A2) { g.s[1] = ResSrc{A2, lda2, G2, Wp2}; g.nsrc = 2; }
        g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.ntiles = hgv_cdiv(M, 16);
        // ring depth: 3 when it divides the tile's step count, else 2 (odd counts get one padding step)
        const int Q = host_quads(G1) + host_quads(G2);
        const bool r3 = Q % 3 == 0;
        int rc = HGV_OK;
#define HGV_RES(n) case n: rc = r3 ? launch_nt_res<n, 3>(g, lds, stream) : launch_nt_res<n, 2>(g, lds, stream); break;
        switch (nt) {
            HGV_RES(1) HGV_RES(2) HGV_RES(4) HGV_RES(7) HGV_RES(9) HGV_RES(12) HGV_RES(16)
            default: rc = r3 ? launch_nt_res<18, 3>(g, lds, stream) : launch_nt_res<18, 2>(g, lds, stream); break;
        }
#undef HGV_RES
        if (rc != HGV_OK) return rc;
        HGV_CHECK_LAUNCH("gemm_nt (resident weights)");
        return HGV_OK;
    }
    GemmArgs g{};
    g.s[0] = GemmSrc{A1, lda1, K1, Wp1};
    g.nsrc = 1;
    if (A2) { g.s[1] = GemmSrc{A2, lda2, K2, Wp2}; g.nsrc = 2; }
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N;
    switch (nt) {
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
