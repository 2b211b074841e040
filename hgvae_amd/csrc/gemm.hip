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

// up to 4 packed images in one launch (the weights of one operator call, both orientations)
struct PackJobs {
    const float* W[4]; int64_t sn[4], sk[4];
    int N[4], K[4], R[4], Gp[4];
    float* out[4];
    int64_t start[5];          // element ranges of the jobs in the flat index space
    int njobs;
};
__global__ __launch_bounds__(256) void k_pack_weights(const PackJobs j) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < j.start[j.njobs]; e += (int64_t)gridDim.x * blockDim.x) {
        int q = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t) q += (t < j.njobs && e >= j.start[t]) ? 1 : 0;
        const int64_t l = e - j.start[q];
        const int R = j.R[q];
        const int gidx = (int)(l / (R * 4)), rem = (int)(l - (int64_t)gidx * R * 4);
        const int n = rem >> 2, k = 4 * gidx + (rem & 3);
        j.out[q][l] = (n < j.N[q] && k < j.K[q]) ? j.W[q][(int64_t)n * j.sn[q] + (int64_t)k * j.sk[q]] : 0.f;
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
        // batches of 8 loads in flight per thread: one load -> store per iteration would pay the L2 latency ~10 times
        auto copy_image = [&](const f32x4* src, f32x4* to, int count) __attribute__((always_inline)) {
            for (int e0 = t; e0 < count; e0 += 8 * blockDim.x) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + u * (int)blockDim.x, count - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * (int)blockDim.x < count) to[e0 + u * (int)blockDim.x] = v[u];
            }
        };
        copy_image(reinterpret_cast<const f32x4*>(g.s[0].Wp), dst, n1);
        if (n2 > 0) copy_image(reinterpret_cast<const f32x4*>(g.s[1].Wp), dst + n1, n2);
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
// TN kernel (weight gradient): operands straight from global memory, no LDS in the main loop
// ------------------------------------------------------------------------------------------
// GW[n][k] = sum_m G[m][n] A[m][k].  One MFMA step reduces 4 rows m: lane (i, kq) supplies row
// m0 + kq.  A lane loads WN consecutive columns of G (per n block) and WK consecutive columns of A
// with one wide load each (16 lanes = 16*WN contiguous floats of a row) and uses element r of a G
// load with element c of the A load in MFMA (r, c) - tile (r, c) therefore holds the output rows
// n = n0 + WN*i' + r and columns k = k0 + WK*j' + c, an interleaving that the reduction kernel
// undoes when it writes the final matrix.  A wave owns NB n blocks x one k block (NB*WN*WK tiles,
// up to 27) of one row slice; its loads run RING steps ahead of the MFMAs.  The kernel is bound by
// the rate at which a compute unit's L1 can be filled from L2, so the tile per wave is made as
// large as the registers allow: G is re-read once per k block, A once per group of n blocks.
// The waves of a workgroup work on consecutive row slices of the same tiles and add their
// accumulators through LDS at the end (a tree, fixed order), which divides the number of partial
// tile sets that go through memory.
//
// A lane whose first column is past the matrix reads column 0 instead; a lane at the edge may read
// up to 3 floats past its row's leading dimension, i.e. the start of the next row.  To keep that
// inside the buffer, and to keep every slice a whole number of 4-row steps, this kernel covers the
// rows [0, Mf) with Mf = 4*floor((M-1)/4); the outer products of the last 1..4 rows are added by
// the reduction kernel.  Columns past the matrix are not masked: they only reach output elements
// that are never stored.
struct TdArgs {
    const float* G; int64_t ldg; int N;
    const float* A; int64_t lda; int K;
    const float* A2; int64_t lda2; int K2;   // optional second source: k blocks [kblk1, kblk) take their columns from A2
    int kblk1;
    int64_t Mf;                 // rows covered here: 4*floor((M-1)/4)
    int64_t rows_per_slice;     // rows per wave, multiple of 4
    int sets;                   // workgroup slices = partial tile sets
    int ngrp, kblk;             // groups of NB n blocks, k blocks
    float* partial;             // [sets][ngrp*kblk][NB*WN*WK][4][64]
    float* bias_partial;        // [sets][ngrp][NB*WN][16] or null
    const float* G2; int64_t ldg2;   // optional second gradient matrix of the same width N (then no A2): the n blocks
    int nblk1n;                      // [nblk1n, 2 nblk1n) take their columns from G2; 0 = one gradient matrix
};

// W floats from a 4-byte aligned address as one load.  The vector has exactly W live elements: a
// dead element of a wider load gets reused as a temporary register, and writing it forces a wait
// for the load that is still in flight.
template <int W> struct TdVec;
template <> struct TdVec<1> { typedef float T; typedef float U __attribute__((aligned(4))); static __device__ __forceinline__ float at(T v, int) { return v; } };
template <> struct TdVec<2> { typedef float T __attribute__((ext_vector_type(2))); typedef float U __attribute__((ext_vector_type(2), aligned(4))); static __device__ __forceinline__ float at(T v, int r) { return v[r]; } };
template <> struct TdVec<3> { typedef float T __attribute__((ext_vector_type(3))); typedef float U __attribute__((ext_vector_type(3), aligned(4))); static __device__ __forceinline__ float at(T v, int r) { return v[r]; } };
template <> struct TdVec<4> { typedef float T __attribute__((ext_vector_type(4))); typedef float U __attribute__((ext_vector_type(4), aligned(4))); static __device__ __forceinline__ float at(T v, int r) { return v[r]; } };

template <int WN, int NB, int WK, int RING>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_direct(const TdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // (waves/2) x (tiles x 256 + NB*WN x 64) floats
    constexpr int TN_ = NB * WN, TILES = TN_ * WK;
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), spw = blockDim.x >> 6;
    // workgroups go round-robin over the 8 XCDs: give every XCD whole row slices, so that the
    // workgroups sharing rows also share an L2
    const int per = g.ngrp * g.kblk;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int set = (idx / per) * 8 + xcd, blk = idx % per;
    if (set >= g.sets) return;                       // whole workgroup: no barrier is skipped by part of it
    const int zb = blk / g.kblk, yb = blk - zb * g.kblk;
    const bool second = yb >= g.kblk1;               // workgroup-uniform
    const int k0 = (second ? yb - g.kblk1 : yb) * 16 * WK + WK * i;
    const int Ks = second ? g.K2 : g.K;
    const int64_t ldas = second ? g.lda2 : g.lda;
    const int64_t m_lo = min(g.Mf, ((int64_t)set * spw + w) * g.rows_per_slice);
    const int64_t m_hi = min(g.Mf, m_lo + g.rows_per_slice);
    // Per-lane byte offsets inside a 4-row step (constant for the whole kernel); the step's base
    // address is wave-uniform, so the per-step address arithmetic is scalar.
    uint32_t goff[NB];
    const char* gbytes[NB];          // wave-uniform: which gradient matrix the n block reads, and its row pitch in bytes
    int64_t gpitch[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int nbi = zb * NB + b;
        const bool g2 = g.nblk1n > 0 && nbi >= g.nblk1n;
        const int n0 = (g2 ? nbi - g.nblk1n : nbi) * 16 * WN + WN * i;
        const int64_t ld = g2 ? g.ldg2 : g.ldg;
        goff[b] = (uint32_t)(((int64_t)kq * ld + (n0 < g.N ? n0 : 0)) * 4);
        gbytes[b] = reinterpret_cast<const char*>(g2 ? g.G2 : g.G);
        gpitch[b] = ld * 4;
    }
    const uint32_t aoff = (uint32_t)(((int64_t)kq * ldas + (k0 < Ks ? k0 : 0)) * 4);
    const char* abytes = reinterpret_cast<const char*>(second ? g.A2 : g.A);

    f32x4 acc[TN_][WK];
#pragma unroll
    for (int r = 0; r < TN_; ++r)
#pragma unroll
        for (int c = 0; c < WK; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[TN_];
#pragma unroll
    for (int r = 0; r < TN_; ++r) bsum[r] = 0.f;

    typename TdVec<WN>::T gr[RING][NB];
    typename TdVec<WK>::T ar[RING];
    int64_t lm = m_lo;           // load cursor (wave-uniform); past the covered rows it re-reads the last step
    auto issue = [&](int d) __attribute__((always_inline)) {
        const int64_t r0 = min(lm, g.Mf - 4);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            gr[d][b] = *reinterpret_cast<const typename TdVec<WN>::U*>(gbytes[b] + r0 * gpitch[b] + goff[b]);
        ar[d] = *reinterpret_cast<const typename TdVec<WK>::U*>(abytes + r0 * ldas * 4 + aoff);
        lm += 4;
    };
    const int steps = (int)((m_hi - m_lo) >> 2);                 // slices hold whole steps
    const int steps_p = (steps + RING - 1) / RING * RING;         // whole ring rounds; the extra steps are masked
#pragma unroll
    for (int d = 0; d < RING; ++d) {
        issue(d);
        __builtin_amdgcn_sched_barrier(0);   // slot order = issue order, or the first wait in the loop drains the ring
    }
    for (int s0 = 0; s0 < steps_p; s0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            typename TdVec<WN>::T gv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) gv[b] = gr[d][b];
            const typename TdVec<WK>::T av = ar[d];
            issue(d);
            const bool live = s0 + d < steps;      // wave-uniform
            float ga[TN_];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < WN; ++r) ga[b * WN + r] = live ? TdVec<WN>::at(gv[b], r) : 0.f;
#pragma unroll
            for (int r = 0; r < TN_; ++r) {
#pragma unroll
                for (int c = 0; c < WK; ++c) acc[r][c] = MFMA16(ga[r], TdVec<WK>::at(av, c), acc[r][c]);
                bsum[r] += ga[r];
            }
            // keep every slot's load where it is written: moved to the end of the unrolled body it
            // would be consumed right after the loop back-edge
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // column sums: add the 4 row groups of the wave
#pragma unroll
    for (int r = 0; r < TN_; ++r) {
        bsum[r] += __shfl_xor(bsum[r], 16, 64);
        bsum[r] += __shfl_xor(bsum[r], 32, 64);
    }
    // tree over the workgroup's waves: the upper half of the active waves writes, the lower half adds
    constexpr int SLOT = TILES * 256 + TN_ * 64;
    for (int half = spw >> 1; half >= 1; half >>= 1) {
        if (w >= half && w < 2 * half) {
            float* dst = red + (int64_t)(w - half) * SLOT;
#pragma unroll
            for (int r = 0; r < TN_; ++r) {
#pragma unroll
                for (int c = 0; c < WK; ++c)
                    *reinterpret_cast<f32x4*>(dst + ((r * WK + c) * 64 + lane) * 4) = acc[r][c];
                dst[TILES * 256 + r * 64 + lane] = bsum[r];
            }
        }
        __syncthreads();
        if (w < half) {
            const float* src = red + (int64_t)w * SLOT;
#pragma unroll
            for (int r = 0; r < TN_; ++r) {
#pragma unroll
                for (int c = 0; c < WK; ++c)
                    acc[r][c] += *reinterpret_cast<const f32x4*>(src + ((r * WK + c) * 64 + lane) * 4);
                bsum[r] += src[TILES * 256 + r * 64 + lane];
            }
        }
        __syncthreads();
    }
    if (w == 0) {
        float* out = g.partial + ((int64_t)set * per + blk) * (TILES * 256);
#pragma unroll
        for (int r = 0; r < TN_; ++r)
#pragma unroll
            for (int c = 0; c < WK; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[((r * WK + c) * 4 + q) * 64 + lane] = acc[r][c][q];
        if (g.bias_partial && yb == 0 && kq == 0) {
            float* bo = g.bias_partial + ((int64_t)set * g.ngrp + zb) * (TN_ * 16);
#pragma unroll
            for (int r = 0; r < TN_; ++r) bo[r * 16 + i] = bsum[r];
        }
    }
}

// Final matrix from the partial tile sets: GW[n][k] = sum over sets (fixed order, fp64) of the
// tile element that holds (n, k), plus the products of the last 1..4 rows; after the tile elements
// come the bias elements (column sums of G).  A workgroup handles 64 elements x 4 groups of sets;
// group sums are combined through LDS in a fixed order, so the result is reproducible.
struct TrArgs {
    const float* partial; const float* bias_partial; int sets;
    int wn, nb, wk, ngrp, kblk;
    const float* g_last; const float* a_last;      // first of the last rows of G and of A (not covered by the tiles)
    int64_t ldg, lda; int n_last;                   // 1..4 such rows
    int N, K;
    float* GW; int64_t ldgw; float* gbias;
    const float* a_last2; int64_t lda2; int K2; int kblk1;   // second source (k blocks >= kblk1) and its output matrix
    float* GW2; int64_t ldgw2;
    int accumulate;                                 // add to GW / gbias instead of overwriting them
    const float* g_last2; int64_t ldg2; int nblk1n; // second gradient matrix (n blocks >= nblk1n): its last rows, its outputs
    float* GWg2; int64_t ldgwg2; float* gbias2;
};

__global__ __launch_bounds__(256) void k_reduce_tn(const TrArgs g) {
    __shared__ double sh[4][65];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int tn = g.nb * g.wn;
    const int64_t tile_elems = (int64_t)g.ngrp * g.kblk * tn * g.wk * 256;
    const int64_t bias_elems = (int64_t)g.ngrp * tn * 16;
    const int64_t e = (int64_t)blockIdx.x * 64 + el;
    const bool is_bias = e >= tile_elems;
    const int64_t eb = e - tile_elems;
    const bool live = !is_bias || (g.gbias != nullptr && eb < bias_elems);
    const float* src = is_bias ? g.bias_partial + eb : g.partial + e;
    const int64_t stride = is_bias ? bias_elems : tile_elems;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (live) {
        int k = grp;
        for (; k + 12 < g.sets; k += 16) {
            s0 += (double)src[(int64_t)k * stride];
            s1 += (double)src[(int64_t)(k + 4) * stride];
            s2 += (double)src[(int64_t)(k + 8) * stride];
            s3 += (double)src[(int64_t)(k + 12) * stride];
        }
        for (; k < g.sets; k += 4) s0 += (double)src[(int64_t)k * stride];
    }
    sh[grp][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && live) {
        double v = ((sh[0][el] + sh[1][el]) + sh[2][el]) + sh[3][el];
        if (is_bias) {
            const int zb = (int)(eb / (tn * 16)), rem = (int)(eb - (int64_t)zb * tn * 16);
            const int rr = rem >> 4, b = rr / g.wn, r = rr - b * g.wn;
            const int nbi = zb * g.nb + b;
            const bool g2 = g.nblk1n > 0 && nbi >= g.nblk1n;
            const int n = (g2 ? nbi - g.nblk1n : nbi) * 16 * g.wn + g.wn * (rem & 15) + r;
            if (n < g.N && !(g.nblk1n > 0 && nbi >= 2 * g.nblk1n)) {
                const float* gl = g2 ? g.g_last2 : g.g_last;
                const int64_t ldl = g2 ? g.ldg2 : g.ldg;
                for (int q = 0; q < g.n_last; ++q) v += (double)gl[q * ldl + n];
                float* dst = (g2 ? g.gbias2 : g.gbias) + n;
                *dst = g.accumulate ? *dst + (float)v : (float)v;
            }
        } else {
            const int per_tile = tn * g.wk * 256;
            const int blk = (int)(e / per_tile), rem = (int)(e - (int64_t)blk * per_tile);
            const int t = rem >> 8, q = (rem >> 6) & 3, lane = rem & 63;
            const int rr = t / g.wk, c = t - rr * g.wk;
            const int b = rr / g.wn, r = rr - b * g.wn;
            const int zb = blk / g.kblk, yb = blk - zb * g.kblk;
            const int nbi = zb * g.nb + b;
            const bool g2 = g.nblk1n > 0 && nbi >= g.nblk1n;
            const int n = (g2 ? nbi - g.nblk1n : nbi) * 16 * g.wn + g.wn * (4 * (lane >> 4) + q) + r;
            const bool second = yb >= g.kblk1;
            const int k = (second ? yb - g.kblk1 : yb) * 16 * g.wk + g.wk * (lane & 15) + c;
            if (n < g.N && k < (second ? g.K2 : g.K) && !(g.nblk1n > 0 && nbi >= 2 * g.nblk1n)) {
                const float* al = second ? g.a_last2 : g.a_last;
                const int64_t ldl = second ? g.lda2 : g.lda;
                const float* gl = g2 ? g.g_last2 : g.g_last;
                const int64_t ldgl = g2 ? g.ldg2 : g.ldg;
                for (int q2 = 0; q2 < g.n_last; ++q2) v += (double)gl[q2 * ldgl + n] * (double)al[q2 * ldl + k];
                float* dst = g2 ? g.GWg2 + (int64_t)n * g.ldgwg2 + k
                                : (second ? g.GW2 + (int64_t)n * g.ldgw2 + k : g.GW + (int64_t)n * g.ldgw + k);
                *dst = g.accumulate ? *dst + (float)v : (float)v;
            }
        }
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

// padded width of a dimension covered by blocks of 16*w columns
int td_pad(int n, int w) { return (int)hgv_cdiv(n, 16 * w) * 16 * w; }

// columns per lane (1..wmax) for a matrix dimension: least padding (ties: wider), at least 2 when
// the dimension has more than one tile
int td_width(int n, int wmax) {
    if (n <= 16 || wmax < 2) return 1;
    int best = 2;
    for (int w = 3; w <= wmax; ++w)
        if (td_pad(n, w) <= td_pad(n, best)) best = w;
    return best;
}

struct TdPlan {
    int wn, nb, wk, ngrp, kblk;      // tile shape per wave and block counts (kblk over both sources)
    int kblk1;                       // k blocks of the first source
    int sets, spw;                   // workgroups along the rows, waves per workgroup
    int64_t rows_per_slice;          // rows per wave
    int64_t tile_elems, bias_elems;  // floats per partial set
};

// nsrc = 2: two gradient matrices of width N side by side (their n blocks follow one another)
TdPlan td_plan(int64_t M, int N, int K, int K2 = 0, int nsrc = 1) {
    TdPlan p{};
    p.wn = td_width(N, 4);
    const int nblk = nsrc * (int)hgv_cdiv(N, 16 * p.wn);
    p.nb = nblk < 3 ? nblk : 3;                               // n blocks per wave
    int wkmax = 27 / (p.nb * p.wn);
    if (wkmax > 4) wkmax = 4;
    if (wkmax < 2 && p.nb > 1) { p.nb = 1; wkmax = 4; }       // very wide N: one block per wave, wide k
    p.wk = td_width(K > K2 ? K : K2, wkmax);
    p.ngrp = (int)hgv_cdiv(nblk, p.nb);
    p.kblk1 = (int)hgv_cdiv(K, 16 * p.wk);
    p.kblk = p.kblk1 + (K2 > 0 ? (int)hgv_cdiv(K2, 16 * p.wk) : 0);
    const int tn = p.nb * p.wn;
    p.tile_elems = (int64_t)p.ngrp * p.kblk * tn * p.wk * 256;
    p.bias_elems = (int64_t)p.ngrp * tn * 16;
    const int64_t Mf = (M - 1) / 4 * 4;
    p.spw = 1; p.rows_per_slice = 4;
    if (Mf <= 0) { p.sets = 0; return p; }
    // two workgroups of four waves per compute unit; every wave gets at least 32 rows
    const int64_t per = (int64_t)p.ngrp * p.kblk;
    int64_t slices = hgv_cdiv(Mf, 32);                        // most row slices the rows allow
    int spw = 4;
    while (spw > 1 && slices < spw) spw >>= 1;
    // workgroups (sets x per, sets rounded up to whole XCD rounds) must not exceed the resident
    // capacity: a few workgroups left over for a second round would double the time
    int64_t sets = (int64_t)2 * cu_count() / per;
    if (sets >= 8) sets = sets / 8 * 8;
    if (sets * spw > slices) sets = slices / spw;
    if (sets < 1) sets = 1;
    p.spw = spw;
    p.rows_per_slice = hgv_cdiv(hgv_cdiv(Mf, sets * spw), 4) * 4;
    p.sets = (int)hgv_cdiv(Mf, p.rows_per_slice * spw);
    return p;
}

template <int WN, int NB, int WK>
int launch_td(const TdArgs& g, int spw, hipStream_t st) {
    constexpr int RING = (NB * WN + WK) * 6 <= 72 ? 6 : 5;
    const int64_t blocks = hgv_cdiv(g.sets, 8) * 8 * g.ngrp * g.kblk;
    size_t lds = (size_t)(spw >> 1) * (NB * WN * WK * 256 + NB * WN * 64) * sizeof(float);
    HGV_REQUIRE(lds <= 64 * 1024, HGV_EUNSUPPORTED, "gemm_tn: reduction buffer of %zu bytes", lds);
    // The dispatcher packs as many workgroups on a compute unit as its resources allow.  Asking
    // for 64 KB of LDS caps that at two, which spreads a grid of 2 x CUs workgroups evenly.
    if (blocks >= cu_count()) lds = 64 * 1024;
    hipLaunchKernelGGL((k_gemm_tn_direct<WN, NB, WK, RING>), dim3((unsigned)blocks), dim3(64 * spw), lds, st, g);
    return HGV_OK;
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

// Up to 4 packed images in one launch: job q packs W[q] (element (n,k) at W[q][n*sn[q] + k*sk[q]], N[q] x K[q])
// into out[q] (hgv_pack_bytes(N[q], K[q]) bytes).  The arrays are host arrays of njobs entries.
extern "C" int hgv_pack_weights(int njobs, const float* const* W, const int64_t* sn, const int64_t* sk, const int* N,
                                const int* K, float* const* out, hipStream_t stream) {
    HGV_REQUIRE(njobs >= 1 && njobs <= 4 && W && sn && sk && N && K && out, HGV_EINVAL, "pack_weights: 1..4 jobs");
    PackJobs j{};
    j.njobs = njobs;
    int64_t total = 0;
    for (int q = 0; q < njobs; ++q) {
        HGV_REQUIRE(W[q] && out[q] && N[q] > 0 && K[q] > 0, HGV_EINVAL, "pack_weights: bad job %d", q);
        HGV_REQUIRE(N[q] <= 288, HGV_EUNSUPPORTED, "pack_weights: N=%d > 288", N[q]);
        HGV_REQUIRE(hgv_aligned(out[q], 16), HGV_EINVAL, "pack_weights: output %d must be 16-byte aligned", q);
        j.W[q] = W[q]; j.sn[q] = sn[q]; j.sk[q] = sk[q]; j.N[q] = N[q]; j.K[q] = K[q];
        j.R[q] = hgv_pack_rows(N[q]); j.Gp[q] = kp_of(K[q]) / 4; j.out[q] = out[q];
        j.start[q] = total;
        total += (int64_t)j.R[q] * j.Gp[q] * 4;
    }
    for (int q = njobs; q <= 4; ++q) j.start[q] = total;
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)hgv_cdiv(total, 256)), dim3(256), 0, stream, j);
    HGV_CHECK_LAUNCH("pack_weights");
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

namespace {

size_t tn_ws_bytes(int64_t M, int N, int K1, int K2, int nsrc = 1) {
    if (M <= 0 || N <= 0 || K1 <= 0) return 256;
    const TdPlan p = td_plan(M, N, K1, K2, nsrc);
    const size_t sets = p.sets > 0 ? p.sets : 1;
    return hgv_align256(sets * p.tile_elems * 4) + hgv_align256(sets * p.bias_elems * 4);
}

int gemm_tn_impl(const float* G, int64_t ldg, int N, const float* A, int64_t lda, int K, float* GW, int64_t ldgw,
                 const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t stream, const float* Gb = nullptr, int64_t ldgb = 0,
                 float* GWb = nullptr, int64_t ldgwb = 0, float* gbias_b = nullptr) {
    const int nsrc = Gb ? 2 : 1;
    if (Gb) HGV_REQUIRE(!A2 && GWb && ldgb >= N && ldgwb >= K && ldgb < (1 << 27) && (gbias == nullptr) == (gbias_b == nullptr),
                        HGV_EINVAL, "gemm_tn_gg: bad second gradient matrix");
    HGV_REQUIRE(M > 0 && N > 0 && K > 0, HGV_EINVAL, "gemm_tn: bad sizes");
    HGV_REQUIRE(G && A && GW && ws && ldg >= N && lda >= K && ldgw >= K, HGV_EINVAL, "gemm_tn: null buffer or leading dimension below the width");
    if (A2) HGV_REQUIRE(K2 > 0 && GW2 && lda2 >= K2 && ldgw2 >= K2 && lda2 < (1 << 27), HGV_EINVAL, "gemm_tn: bad second source");
    else K2 = 0;
    HGV_REQUIRE(N <= 288, HGV_EUNSUPPORTED, "gemm_tn: N=%d > 288", N);
    HGV_REQUIRE(ldg < (1 << 27) && lda < (1 << 27), HGV_EUNSUPPORTED, "gemm_tn: leading dimension too large");
    HGV_REQUIRE(ws_bytes >= tn_ws_bytes(M, N, K, K2, nsrc) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "gemm_tn: workspace too small or misaligned");
    const TdPlan p = td_plan(M, N, K, K2, nsrc);
    const int nblk1n = Gb ? (int)hgv_cdiv(N, 16 * p.wn) : 0;
    const size_t sets = p.sets > 0 ? p.sets : 1;
    float* partial = (float*)ws;
    float* bias_partial = (float*)((char*)ws + hgv_align256(sets * p.tile_elems * 4));
    const int64_t Mf = (M - 1) / 4 * 4;
    if (p.sets > 0) {
        TdArgs g{};
        g.G = G; g.ldg = ldg; g.N = N; g.A = A; g.lda = lda; g.K = K; g.Mf = Mf;
        g.A2 = A2; g.lda2 = lda2; g.K2 = K2; g.kblk1 = p.kblk1;
        g.rows_per_slice = p.rows_per_slice; g.sets = p.sets; g.ngrp = p.ngrp; g.kblk = p.kblk;
        g.partial = partial; g.bias_partial = gbias ? bias_partial : nullptr;
        g.G2 = Gb; g.ldg2 = ldgb; g.nblk1n = nblk1n;
        int rc = HGV_EUNSUPPORTED;
        bool hit = false;
#define HGV_TD(a, b, c) if (!hit && p.wn == a && p.nb == b && p.wk == c) { hit = true; rc = launch_td<a, b, c>(g, p.spw, stream); }
        HGV_TD(1, 1, 1) HGV_TD(1, 1, 2) HGV_TD(1, 1, 3) HGV_TD(1, 1, 4)
        HGV_TD(1, 2, 1) HGV_TD(1, 2, 2) HGV_TD(1, 2, 3) HGV_TD(1, 2, 4)      // two narrow gradient matrices side by side
        HGV_TD(2, 1, 1) HGV_TD(2, 1, 2) HGV_TD(2, 1, 3) HGV_TD(2, 1, 4)
        HGV_TD(2, 2, 1) HGV_TD(2, 2, 2) HGV_TD(2, 2, 3) HGV_TD(2, 2, 4)
        HGV_TD(2, 3, 1) HGV_TD(2, 3, 2) HGV_TD(2, 3, 3) HGV_TD(2, 3, 4)
        HGV_TD(3, 1, 1) HGV_TD(3, 1, 2) HGV_TD(3, 1, 3) HGV_TD(3, 1, 4)
        HGV_TD(3, 2, 1) HGV_TD(3, 2, 2) HGV_TD(3, 2, 3) HGV_TD(3, 2, 4)
        HGV_TD(3, 3, 1) HGV_TD(3, 3, 2) HGV_TD(3, 3, 3)
        HGV_TD(4, 1, 1) HGV_TD(4, 1, 2) HGV_TD(4, 1, 3) HGV_TD(4, 1, 4)
        HGV_TD(4, 2, 1) HGV_TD(4, 2, 2) HGV_TD(4, 2, 3)
        HGV_TD(4, 3, 1) HGV_TD(4, 3, 2)
#undef HGV_TD
        HGV_REQUIRE(hit, HGV_EUNSUPPORTED, "gemm_tn: no kernel for tile shape (%d,%d,%d)", p.wn, p.nb, p.wk);
        if (rc != HGV_OK) return rc;
    }
    TrArgs r{};
    r.partial = partial; r.bias_partial = bias_partial; r.sets = p.sets;
    r.wn = p.wn; r.nb = p.nb; r.wk = p.wk; r.ngrp = p.ngrp; r.kblk = p.kblk;
    r.g_last = G + Mf * ldg; r.a_last = A + Mf * lda; r.ldg = ldg; r.lda = lda; r.n_last = (int)(M - Mf);
    r.N = N; r.K = K; r.GW = GW; r.ldgw = ldgw; r.gbias = gbias; r.accumulate = accumulate;
    r.a_last2 = A2 ? A2 + Mf * lda2 : nullptr; r.lda2 = lda2; r.K2 = K2; r.kblk1 = p.kblk1; r.GW2 = GW2; r.ldgw2 = ldgw2;
    r.g_last2 = Gb ? Gb + Mf * ldgb : nullptr; r.ldg2 = ldgb; r.nblk1n = nblk1n; r.GWg2 = GWb; r.ldgwg2 = ldgwb; r.gbias2 = gbias_b;
    const int64_t elems = p.tile_elems + (gbias ? p.bias_elems : 0);
    hipLaunchKernelGGL(k_reduce_tn, dim3((unsigned)hgv_cdiv(elems, 64)), dim3(256), 0, stream, r);
    HGV_CHECK_LAUNCH("gemm_tn");
    return HGV_OK;
}

}  // namespace

extern "C" size_t hgv_gemm_tn_ws(int64_t M, int N, int K) { return tn_ws_bytes(M, N, K, 0); }
extern "C" size_t hgv_gemm_tn2_ws(int64_t M, int N, int K1, int K2) { return tn_ws_bytes(M, N, K1, K2); }
extern "C" size_t hgv_gemm_tn_gg_ws(int64_t M, int N, int K) { return tn_ws_bytes(M, N, K, 0, 2); }

// GW[N,K] (row-major, leading dimension ldgw >= K) (+)= G[M,N]^T A[M,K];  gbias[N] (+)= column sums of G (nullable).
// accumulate != 0 adds to GW / gbias (gradient accumulation into an existing .grad buffer) instead of overwriting.
// G and A are row-major with leading dimensions >= their widths (any 4-byte aligned pointers).
extern "C" int hgv_gemm_tn(const float* G, int64_t ldg, int N, const float* A, int64_t lda, int K, int64_t M,
                           float* GW, int64_t ldgw, float* gbias, int accumulate, void* ws, size_t ws_bytes,
                           hipStream_t stream) {
    return gemm_tn_impl(G, ldg, N, A, lda, K, GW, ldgw, nullptr, 0, 0, nullptr, 0, M, gbias, accumulate, ws, ws_bytes, stream);
}

// Two weight gradients that share G in one pass: GW1[N,K1] (+)= G^T A1 and GW2[N,K2] (+)= G^T A2 (the two halves of
// a K-concatenated projection; GW1 / GW2 may be column blocks of one matrix); gbias as in hgv_gemm_tn.
extern "C" int hgv_gemm_tn2(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                            const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(A2 && GW2, HGV_EINVAL, "gemm_tn2: null second source");
    return gemm_tn_impl(G, ldg, N, A1, lda1, K1, GW1, ldgw1, A2, lda2, K2, GW2, ldgw2, M, gbias, accumulate, ws, ws_bytes, stream);
}

// Two weight gradients that share A in one pass: GWa[N,K] (+)= Ga^T A and GWb[N,K] (+)= Gb^T A for two gradient matrices
// of the same width (a projection applied before the aggregation: W_l meets the transposed aggregate of the output
// gradient, W_r the output gradient itself, both against the same input); gbias_a / gbias_b (+)= their column sums
// (both or neither).
extern "C" int hgv_gemm_tn_gg(const float* Ga, int64_t ldga, const float* Gb, int64_t ldgb, int N, const float* A, int64_t lda,
                              int K, int64_t M, float* GWa, int64_t ldgwa, float* GWb, int64_t ldgwb, float* gbias_a,
                              float* gbias_b, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(Gb && GWb, HGV_EINVAL, "gemm_tn_gg: null second gradient matrix");
    HGV_REQUIRE(2 * N <= 288, HGV_EUNSUPPORTED, "gemm_tn_gg: 2N=%d > 288", 2 * N);
    return gemm_tn_impl(Ga, ldga, N, A, lda, K, GWa, ldgwa, nullptr, 0, 0, nullptr, 0, M, gbias_a, accumulate, ws, ws_bytes, stream,
                        Gb, ldgb, GWb, ldgwb, gbias_b);
}
