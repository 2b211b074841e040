// Fused (ReLU) -> Dropout -> BatchNorm1d over the node dimension, forward and backward.
//
// Replaces the three-module tail of every encoder/decoder level of the reference
// (gvae/gvae_skipcon.py:105-107,110-112,115-117,124,130,136;
// gvae/gvae_skipcon_attention.py:104-115; autoencoder/vae_with_pool.py:96-103,109-111):
//   d   = dropout(relu(y))            (relu optional; mask explicit, Philox, or none)
//   out = (d - mean_c) * rstd_c * gamma_c + beta_c,   mean/var over the N rows (biased var)
//   running stats: momentum update with the unbiased variance.
//
// The activation is never materialised: the statistics pass and the normalise pass both
// recompute d from the pre-activation y (3 N*H floats of traffic instead of 5).
// Column statistics: each workgroup reduces its rows (ep_block_rows) with sums shifted by the mean of its
// first rows (no catastrophic cancellation), writes (mean_b, M2_b); a second tiny kernel
// merges the partials with Chan's formula in fp64.  No atomics -> bitwise reproducible.
//
// Thread mapping: a thread owns the 4-column chunks cp and cp + CP of a row (CP = half the chunks, so
// consecutive threads still read consecutive 16-byte pieces); 256/CP row slots run side by side.
// One Philox call yields the 8 keep decisions of a chunk pair (16 random bits per element, i.e.
// keep probability quantised to 1/65536): the integer multiplies of the generator, not memory,
// bounded these kernels when every 4 elements paid for a call of their own.
#include "common.h"
#include <stdlib.h>
#include <mutex>

namespace {

struct EpArgs {
    const float* y; int64_t ldy;
    int64_t N; int H; int CH; int CP; int RPB;   // CH chunks, CP chunk pairs per row, RPB rows side by side per workgroup
    int relu; int mode;              // mode 0: no dropout, 1: explicit mask, 2: Philox
    float p; float scale; uint32_t thresh;   // Philox: keep iff 16 random bits >= thresh = round(p * 65536)
    const uint8_t* mask;             // [N, H] contiguous
    uint64_t seed, offset;
    uint64_t offset2;                // Philox offset of the second noise set (kernels instantiated for two sets)
    const uint64_t* offset_dev;      // optional device word added to offset (advanced once per step under graph replay)
    int pre;                         // y already holds d = dropout(relu(y)) (written by the producing kernel; without a relu
                                     // a dropped element is -0.0f, a kept zero +0.0f): nothing is recomputed or regenerated
    const float* y2;                 // pre, two noise sets: the activated matrix of the second set (same leading dimension as y)
};

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* base, int64_t ld, int64_t row, int c0, int H) {
    const float* p = base + row * ld + c0;
    float4 v;
    if (VEC) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = c0 + 0 < H ? p[0] : 0.f; v.y = c0 + 1 < H ? p[1] : 0.f;
        v.z = c0 + 2 < H ? p[2] : 0.f; v.w = c0 + 3 < H ? p[3] : 0.f;
    }
    return v;
}
template <bool VEC>
__device__ __forceinline__ void st4(float* base, int64_t ld, int64_t row, int c0, int H, float4 v) {
    float* p = base + row * ld + c0;
    if (VEC) {
        if (c0 + 1 >= H) v.y = 0.f;
        if (c0 + 2 >= H) v.z = 0.f;
        if (c0 + 3 >= H) v.w = 0.f;
        *reinterpret_cast<float4*>(p) = v;
    } else {
        if (c0 + 0 < H) p[0] = v.x;
        if (c0 + 1 < H) p[1] = v.y;
        if (c0 + 2 < H) p[2] = v.z;
        if (c0 + 3 < H) p[3] = v.w;
    }
}

// A dropout rate that is a multiple of 1 / 256 (the reference's 0.5 is) needs one byte of the generator per decision instead of
// sixteen bits: one call then serves TWO chunk pairs - call q of a row gives pair 2q the bytes of its words x (chunk 2q) and y
// (chunk 2q + CP), pair 2q + 1 those of z and w.  Kernels whose threads own one pair gain nothing (each thread makes the call of
// its pair's even partner); the kernel that only packs decisions into bits makes half the calls.
__device__ __forceinline__ bool keep_by_bytes(const EpArgs& a) { return (a.thresh & 0xffu) == 0u; }
__device__ __forceinline__ uint4 keep_bytes16(const EpArgs& a, const HgvPhilox& ph, int64_t row, int q, uint64_t off) {
    return ph.rounds<7>((uint32_t)row, (uint32_t)((uint64_t)row >> 32) ^ ((uint32_t)q << 8) ^ 0x40000000u, (uint32_t)off,
                        (uint32_t)(off >> 32));
}

// keep-multipliers (0 or scale) for the two 4-column chunks (cp, cp + CP) of chunk pair cp
__device__ __forceinline__ void keep8(const EpArgs& a, int64_t row, int cp, float4 (&k)[2], int set = 0) {
    k[0] = k[1] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.mode == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = (cp + h * a.CP) * 4;
            const uint8_t* m = a.mask + row * a.H + c0;
            k[h].x = (c0 + 0 < a.H && m[0]) ? a.scale : 0.f;
            k[h].y = (c0 + 1 < a.H && m[1]) ? a.scale : 0.f;
            k[h].z = (c0 + 2 < a.H && m[2]) ? a.scale : 0.f;
            k[h].w = (c0 + 3 < a.H && m[3]) ? a.scale : 0.f;
        }
    } else if (a.mode == 2) {
        const HgvPhilox ph(a.seed);
        const uint64_t off = (set ? a.offset2 : a.offset) + (a.offset_dev ? a.offset_dev[0] : 0ull);
        if (keep_by_bytes(a)) {
            // one byte per decision: the call of the pair's even partner serves both pairs (see keep_bytes16)
            const uint4 r = keep_bytes16(a, ph, row, cp >> 1, off);
            const uint32_t T8 = a.thresh >> 8, w0 = (cp & 1) ? r.z : r.x, w1 = (cp & 1) ? r.w : r.y;
            k[0].x = (w0 & 0xffu) >= T8 ? a.scale : 0.f; k[0].y = ((w0 >> 8) & 0xffu) >= T8 ? a.scale : 0.f;
            k[0].z = ((w0 >> 16) & 0xffu) >= T8 ? a.scale : 0.f; k[0].w = (w0 >> 24) >= T8 ? a.scale : 0.f;
            k[1].x = (w1 & 0xffu) >= T8 ? a.scale : 0.f; k[1].y = ((w1 >> 8) & 0xffu) >= T8 ? a.scale : 0.f;
            k[1].z = ((w1 >> 16) & 0xffu) >= T8 ? a.scale : 0.f; k[1].w = (w1 >> 24) >= T8 ? a.scale : 0.f;
            return;
        }
        const uint4 r = ph.rounds<7>((uint32_t)row, (uint32_t)((uint64_t)row >> 32) ^ ((uint32_t)cp << 8),
                                     (uint32_t)off, (uint32_t)(off >> 32));
        const uint32_t T = a.thresh;
        k[0].x = (r.x & 0xffffu) >= T ? a.scale : 0.f; k[0].y = (r.x >> 16) >= T ? a.scale : 0.f;
        k[0].z = (r.y & 0xffffu) >= T ? a.scale : 0.f; k[0].w = (r.y >> 16) >= T ? a.scale : 0.f;
        k[1].x = (r.z & 0xffffu) >= T ? a.scale : 0.f; k[1].y = (r.z >> 16) >= T ? a.scale : 0.f;
        k[1].z = (r.w & 0xffffu) >= T ? a.scale : 0.f; k[1].w = (r.w >> 16) >= T ? a.scale : 0.f;
    }
}

// Column offsets of the thread's two chunks; a missing second chunk (odd chunk count) aliases the
// first one, so every load stays unconditional and in bounds - its results are simply not used.
__device__ __forceinline__ void chunk_cols(const EpArgs& a, int cp, int (&c)[2]) {
    c[0] = cp * 4;
    c[1] = (cp + a.CP < a.CH ? cp + a.CP : cp) * 4;
}

// d = dropout(relu(y)) for the two chunks of pair cp in one row; yv holds the loaded y values
__device__ __forceinline__ float keep_of(const EpArgs& a, float v) {
    // activated input: behind a relu the derivative is zero wherever the value is (dropped or clamped alike); without one
    // the dropped elements are the negative zeros
    if (a.relu) return a.scale;
    return __float_as_uint(v) == 0x80000000u ? 0.f : a.scale;
}

__device__ __forceinline__ void apply_d2(const EpArgs& a, int64_t row, int cp, const float4 (&yv)[2], float4 (&d)[2],
                                         float4* keep_out = nullptr, int set = 0) {
    if (a.pre) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d[h] = yv[h];
            if (keep_out) keep_out[h] = make_float4(keep_of(a, yv[h].x), keep_of(a, yv[h].y), keep_of(a, yv[h].z), keep_of(a, yv[h].w));
        }
        return;
    }
    float4 k[2];
    if (a.mode != 0) keep8(a, row, cp, k, set);
    else k[0] = k[1] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 v = yv[h];
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.mode != 0) { v.x *= k[h].x; v.y *= k[h].y; v.z *= k[h].z; v.w *= k[h].w; }
        if (keep_out) keep_out[h] = k[h];
        d[h] = v;
    }
}

constexpr int EP_PIVOT_ROWS = 8;
constexpr int EP_WGS_TWO_SETS = 768;       // grid cap of the two-set backward pass: three workgroups per CU are resident

// Rows of a workgroup: the groups of RPB consecutive rows number b, b + G, b + 2G, ... (G workgroups in the launch), so that at
// any moment the launch as a whole works inside one window of G * RPB rows of every matrix it streams - neighbouring
// workgroups share DRAM pages instead of each opening its own (worth 2 - 5 % per launch at 1.3 M rows against a contiguous
// row range per workgroup).  The partial sums of a workgroup are over this fixed set, so results do not depend on scheduling.
__host__ __device__ __forceinline__ int64_t ep_block_rows(int b, int G, int RPB, int64_t N) {
    const int64_t NG = (N + RPB - 1) / RPB;                       // row groups in all; only the last one may be short
    if (b >= NG) return 0;
    const int64_t groups = (NG - 1 - b) / G + 1;
    return groups * RPB - ((NG - 1) % G == b ? NG * RPB - N : 0);
}

// partial layout: [set][block][2][Hp] = (mean_b, M2_b), Hp = CH*4.  NS = 2: two noise sets (Philox offsets a.offset and
// a.offset2) over one read of y - the statistics of two dropout draws of the same pre-activation.
// WD: the activated values d are also written back over y (one set only): a dropped element as -0.0f when there is no relu
// in front (a kept zero as +0.0f), so that later passes read relu and dropout off the stored values (EpArgs::pre).
__device__ __forceinline__ float stored_d(const EpArgs& a, float d, float k) {
    if (a.mode != 0 && k == 0.f) return a.relu ? 0.f : -0.f;
    return d == 0.f ? 0.f : d;
}

template <bool VEC, int NS, bool WD = false>
__global__ __launch_bounds__(256, (NS == 2 && WD) ? 4 : 1) void k_ep_stats(const EpArgs a, float* __restrict__ partial, float* __restrict__ d_out = nullptr,
                                                  float* __restrict__ d_out2 = nullptr) {
    __shared__ float4 sh[512];
    __shared__ float4 sh2[512];
    const int t = threadIdx.x;
    const bool active = t < a.RPB * a.CP;
    const int rsub = t / a.CP, cp = t - rsub * a.CP;
    const int nh = cp + a.CP < a.CH ? 2 : 1;
    const int64_t r0 = (int64_t)blockIdx.x * a.RPB, r1 = a.N, rstep = (int64_t)gridDim.x * a.RPB;       // see ep_block_rows
    int cc[2];
    chunk_cols(a, cp, cc);
    // shift of the sums: the column mean over the block's first EP_PIVOT_ROWS rows.  A single row as the shift can sit
    // several standard deviations from the mean, and S2 - S1^2/n then loses that factor squared in fp32 - visible when
    // a batch has only a few hundred rows in all; any shift within about one standard deviation is as good as the mean.
    float4 s[NS][2], S1[NS][2], S2[NS][2];
#pragma unroll
    for (int set = 0; set < NS; ++set) {
        s[set][0] = s[set][1] = S1[set][0] = S1[set][1] = S2[set][0] = S2[set][1] = make_float4(0, 0, 0, 0);
        float4 d[2];
        d[0] = d[1] = make_float4(0, 0, 0, 0);
        if (active && rsub < EP_PIVOT_ROWS && r0 + rsub < r1) {
            float4 yv[2];
            yv[0] = ld4<VEC>(a.y, a.ldy, r0 + rsub, cc[0], a.H); yv[1] = ld4<VEC>(a.y, a.ldy, r0 + rsub, cc[1], a.H);
            apply_d2(a, r0 + rsub, cp, yv, d, nullptr, set);
        }
        if (set > 0) __syncthreads();               // the previous set's pivot rows have been read
        if (rsub < EP_PIVOT_ROWS) { sh[2 * t] = d[0]; sh[2 * t + 1] = d[1]; }
        __syncthreads();
        if (active) {
            const int cnt = (int)min((int64_t)min(a.RPB, EP_PIVOT_ROWS), r1 - r0);
            for (int r = 0; r < cnt; ++r) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 u = sh[2 * (r * a.CP + cp) + h];
                    s[set][h].x += u.x; s[set][h].y += u.y; s[set][h].z += u.z; s[set][h].w += u.w;
                }
            }
            const float inv = 1.f / (float)cnt;
#pragma unroll
            for (int h = 0; h < 2; ++h) { s[set][h].x *= inv; s[set][h].y *= inv; s[set][h].z *= inv; s[set][h].w *= inv; }
        }
    }
    if (active) {
        for (int64_t row = r0 + rsub; row < r1; row += rstep) {
            float4 yv[2];
            yv[0] = ld4<VEC>(a.y, a.ldy, row, cc[0], a.H); yv[1] = ld4<VEC>(a.y, a.ldy, row, cc[1], a.H);
#pragma unroll
            for (int set = 0; set < NS; ++set) {
                float4 d[2], kk[2];
                apply_d2(a, row, cp, yv, d, WD ? kk : nullptr, set);
                if (WD) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (h < nh) {
                            const float4 w = make_float4(stored_d(a, d[h].x, kk[h].x), stored_d(a, d[h].y, kk[h].y),
                                                         stored_d(a, d[h].z, kk[h].z), stored_d(a, d[h].w, kk[h].w));
                            st4<VEC>(set ? d_out2 : d_out, a.ldy, row, cc[h], a.H, w);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float tx = d[h].x - s[set][h].x, ty = d[h].y - s[set][h].y, tz = d[h].z - s[set][h].z, tw = d[h].w - s[set][h].w;
                    S1[set][h].x += tx; S1[set][h].y += ty; S1[set][h].z += tz; S1[set][h].w += tw;
                    S2[set][h].x += tx * tx; S2[set][h].y += ty * ty; S2[set][h].z += tz * tz; S2[set][h].w += tw * tw;
                }
            }
        }
    }
    const int Hp = a.CH * 4;
#pragma unroll
    for (int set = 0; set < NS; ++set) {
        __syncthreads();
        sh[2 * t] = S1[set][0]; sh[2 * t + 1] = S1[set][1]; sh2[2 * t] = S2[set][0]; sh2[2 * t + 1] = S2[set][1];
        __syncthreads();
        if (active && rsub == 0) {
            // the row slots' sums are added in fp64 (up to 256 of them for narrow rows), and so is M2 = S2 - S1^2 / n
            double D1[2][4], D2[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                D1[h][0] = S1[set][h].x; D1[h][1] = S1[set][h].y; D1[h][2] = S1[set][h].z; D1[h][3] = S1[set][h].w;
                D2[h][0] = S2[set][h].x; D2[h][1] = S2[set][h].y; D2[h][2] = S2[set][h].z; D2[h][3] = S2[set][h].w;
            }
            for (int r = 1; r < a.RPB; ++r) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 u = sh[2 * (r * a.CP + cp) + h], w = sh2[2 * (r * a.CP + cp) + h];
                    D1[h][0] += u.x; D1[h][1] += u.y; D1[h][2] += u.z; D1[h][3] += u.w;
                    D2[h][0] += w.x; D2[h][1] += w.y; D2[h][2] += w.z; D2[h][3] += w.w;
                }
            }
            const double n = (double)ep_block_rows(blockIdx.x, gridDim.x, a.RPB, a.N);
            float* dst = partial + (int64_t)set * gridDim.x * 2 * Hp;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) break;
                const float sv[4] = {s[set][h].x, s[set][h].y, s[set][h].z, s[set][h].w};
                float mv[4], qv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mv[e] = (float)((double)sv[e] + D1[h][e] / n);
                    qv[e] = (float)(D2[h][e] - D1[h][e] * D1[h][e] / n);
                }
                const int c0 = (cp + h * a.CP) * 4;
                *reinterpret_cast<float4*>(dst + ((int64_t)blockIdx.x * 2 + 0) * Hp + c0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                *reinterpret_cast<float4*>(dst + ((int64_t)blockIdx.x * 2 + 1) * Hp + c0) = make_float4(qv[0], qv[1], qv[2], qv[3]);
            }
        }
    }
}

// One workgroup = 16 columns x 64 slices of the partial list; slices are combined through LDS
// in a fixed order (64 -> 8 -> 1), so the result does not depend on scheduling.
constexpr int FIN_SL = 64;

// sums the FIN_SL per-slice values of column cl; valid in the threads with sl == 0
__device__ __forceinline__ double fin_tree(double (*sh)[17], int sl, int cl, double v) {
    sh[sl][cl] = v;
    __syncthreads();
    double s = 0.0;
    if (sl < 8) {
#pragma unroll
        for (int q = 0; q < FIN_SL / 8; ++q) s += sh[sl + 8 * q][cl];
    }
    __syncthreads();
    if (sl < 8) sh[sl][cl] = s;
    __syncthreads();
    s = 0.0;
    if (sl == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s += sh[q][cl];
    }
    __syncthreads();
    return s;
}

// Per-block (mean_b, M2_b) -> batch mean and variance in one pass over the partial list: with the
// pivot p = mean of block 0 and d_b = mean_b - p,
//   mean = p + sum(n_b d_b) / N,   M2 = sum(M2_b + n_b d_b^2) - (sum n_b d_b)^2 / N      (fp64).
__global__ __launch_bounds__(1024) void k_ep_stats_finalize(const float* __restrict__ partial, int nblocks,
                                                           int RPB, int64_t N, int H, int Hp,
                                                           float eps, float momentum,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd,
                                                           int64_t* __restrict__ num_batches_tracked, int nsets,
                                                           const float* __restrict__ gamma = nullptr,
                                                           const float* __restrict__ beta = nullptr,
                                                           float* __restrict__ col_s = nullptr,
                                                           float* __restrict__ col_t = nullptr) {
    __shared__ double sh[FIN_SL][17];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += nsets;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < H;
    // sets one after the other: the running statistics take the momentum update of set 0, then that of set 1, as two
    // forward calls would apply them; save_mean / save_rstd of set s start at element s * H
    for (int set = 0; set < nsets; ++set) {
        const float* part = partial + (int64_t)set * nblocks * 2 * Hp;
        double a1 = 0.0, a2 = 0.0, pivot = 0.0;
        if (ok) {
            pivot = (double)part[c];
            // batches of 8 independent loads (clamped index, zero weight past the end) before the fp64 chain
            for (int b0 = sl; b0 < nblocks; b0 += 8 * FIN_SL) {
                float m[8], q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = min(b0 + u * FIN_SL, nblocks - 1);
                    m[u] = part[((int64_t)b * 2) * Hp + c];
                    q[u] = part[((int64_t)b * 2 + 1) * Hp + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + u * FIN_SL;
                    const double nb = b >= nblocks ? 0.0 : (double)ep_block_rows(b, nblocks, RPB, N);
                    const double d = (double)m[u] - pivot;
                    a1 += nb * d;
                    a2 += (b < nblocks ? (double)q[u] : 0.0) + nb * d * d;
                }
            }
        }
        const double s1 = fin_tree(sh, sl, cl, a1);
        const double s2 = fin_tree(sh, sl, cl, a2);
        if (sl == 0 && !ok && col_s && c < Hp) col_s[set * Hp + c] = col_t[set * Hp + c] = 0.f;       // pad columns stay zero
        if (sl == 0 && ok) {
            const double mean = pivot + s1 / (double)N;
            double m2 = s2 - s1 * s1 / (double)N;
            if (m2 < 0.0) m2 = 0.0;
            const double var = m2 / (double)N;
            save_mean[set * H + c] = (float)mean;
            save_rstd[set * H + c] = (float)(1.0 / sqrt(var + (double)eps));
            if (col_s) {
                // the normalisation as one multiply-add per element: out = d * s + t with the rounded rstd the backward uses
                const double sc = (double)(gamma ? gamma[c] : 1.f) * (double)save_rstd[set * H + c];
                col_s[set * Hp + c] = (float)sc;
                col_t[set * Hp + c] = (float)((double)(beta ? beta[c] : 0.f) - mean * (double)(float)sc);
            }
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            if (running_var) {
                const double unb = N > 1 ? m2 / (double)(N - 1) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
}

struct NormArgs {
    const float* mean; const float* rstd_or_var; int rstd_is_var; float eps;
    const float* gamma; const float* beta;
    float* out; int64_t ldo;
    float* out2;                     // two noise sets: output of the second set (same leading dimension); its statistics
                                     // follow the first set's at element H of mean / rstd_or_var
};

__device__ __forceinline__ float4 ldcol4(const float* p, int c0, int H, float fill) {
    float4 v;
    v.x = (p && c0 + 0 < H) ? p[c0 + 0] : fill; v.y = (p && c0 + 1 < H) ? p[c0 + 1] : fill;
    v.z = (p && c0 + 2 < H) ? p[c0 + 2] : fill; v.w = (p && c0 + 3 < H) ? p[c0 + 3] : fill;
    return v;
}

template <bool VEC, int NS>
__global__ __launch_bounds__(256) void k_ep_apply(const EpArgs a, const NormArgs n) {
    const int t = threadIdx.x;
    if (t >= a.RPB * a.CP) return;
    const int rsub = t / a.CP, cp = t - rsub * a.CP;
    const int nh = cp + a.CP < a.CH ? 2 : 1;
    const int64_t r0 = (int64_t)blockIdx.x * a.RPB, r1 = a.N, rstep = (int64_t)gridDim.x * a.RPB;       // see ep_block_rows
    float4 mu[NS][2], rs[NS][2], ga[2], be[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (cp + h * a.CP) * 4;
#pragma unroll
        for (int set = 0; set < NS; ++set) {
            mu[set][h] = ldcol4(n.mean ? n.mean + set * a.H : nullptr, c0, a.H, 0.f);
            rs[set][h] = ldcol4(n.rstd_or_var ? n.rstd_or_var + set * a.H : nullptr, c0, a.H, 1.f);
            if (n.rstd_is_var) {
                rs[set][h].x = 1.f / sqrtf(rs[set][h].x + n.eps); rs[set][h].y = 1.f / sqrtf(rs[set][h].y + n.eps);
                rs[set][h].z = 1.f / sqrtf(rs[set][h].z + n.eps); rs[set][h].w = 1.f / sqrtf(rs[set][h].w + n.eps);
            }
        }
        ga[h] = ldcol4(n.gamma, c0, a.H, 1.f); be[h] = ldcol4(n.beta, c0, a.H, 0.f);
    }
    int cc[2];
    chunk_cols(a, cp, cc);
    for (int64_t row = r0 + rsub; row < r1; row += rstep) {
        float4 yv[2];
        yv[0] = ld4<VEC>(a.y, a.ldy, row, cc[0], a.H); yv[1] = ld4<VEC>(a.y, a.ldy, row, cc[1], a.H);
#pragma unroll
        for (int set = 0; set < NS; ++set) {
            float4 d[2];
            apply_d2(a, row, cp, yv, d, nullptr, set);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h < nh) {
                    float4 o;
                    o.x = (d[h].x - mu[set][h].x) * rs[set][h].x * ga[h].x + be[h].x; o.y = (d[h].y - mu[set][h].y) * rs[set][h].y * ga[h].y + be[h].y;
                    o.z = (d[h].z - mu[set][h].z) * rs[set][h].z * ga[h].z + be[h].z; o.w = (d[h].w - mu[set][h].w) * rs[set][h].w * ga[h].w + be[h].w;
                    st4<VEC>(set ? n.out2 : n.out, n.ldo, row, (cp + h * a.CP) * 4, a.H, o);
                }
            }
        }
    }
}

// ---------------- backward ----------------
struct BwdArgs {
    const float* g; int64_t ldg;
    const float* g2; int64_t ldg2;   // optional second incoming gradient (the output had two consumers): g + g2 is used
    const float* mean; const float* rstd; const float* gamma;
    const float* sums;   // [set][2][Hp]: sum g, sum g*xhat (apply pass only)
    float* dy; int64_t lddy;
    // two noise sets: the incoming gradients of the second set's output (h2 nullable); its mean / rstd follow the
    // first set's at element H; dy receives the sum of the two sets' input gradients
    const float* h; int64_t ldh;
    const float* h2; int64_t ldh2;
};

// incoming gradient chunk of a set: g (+ g2 when the forward output fed two consumers)
template <bool VEC>
__device__ __forceinline__ float4 ldg4(const BwdArgs& b, int64_t row, int c0, int H, int set = 0) {
    const float* p = set ? b.h : b.g; const int64_t ld = set ? b.ldh : b.ldg;
    const float* p2 = set ? b.h2 : b.g2; const int64_t ld2 = set ? b.ldh2 : b.ldg2;
    float4 v = ld4<VEC>(p, ld, row, c0, H);
    if (p2) {
        const float4 w = ld4<VEC>(p2, ld2, row, c0, H);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}

// partial layout: [set][block][2][Hp] = (sum g, sum g*xhat)
template <bool VEC, int NS>
__global__ __launch_bounds__(256, 4) void k_ep_bwd_stats(const EpArgs a, const BwdArgs b, float* __restrict__ partial) {
    __shared__ float4 sh[512];
    __shared__ float4 sh2[512];
    const int t = threadIdx.x;
    const bool active = t < a.RPB * a.CP;
    const int rsub = t / a.CP, cp = t - rsub * a.CP;
    const int nh = cp + a.CP < a.CH ? 2 : 1;
    const int64_t r0 = (int64_t)blockIdx.x * a.RPB, r1 = a.N, rstep = (int64_t)gridDim.x * a.RPB;       // see ep_block_rows
    float4 S1[NS][2], S2[NS][2];
#pragma unroll
    for (int set = 0; set < NS; ++set) S1[set][0] = S1[set][1] = S2[set][0] = S2[set][1] = make_float4(0, 0, 0, 0);
    // NS = 2: mean and rstd of both sets sit in LDS ([set][2][Hp], in sh2 until the sums need it) and are fetched inside
    // the row loop; as registers they cost the fourth workgroup per CU (see k_ep_bwd_apply)
    float* cst = reinterpret_cast<float*>(sh2);
    if (NS == 2) {
        const int Hp = a.CH * 4;
        for (int i = t; i < 2 * 2 * Hp; i += 256) {
            const int set = i / (2 * Hp), arr = (i / Hp) & 1, c = i % Hp;
            cst[i] = c < a.H ? (arr ? b.rstd : b.mean)[set * a.H + c] : 0.f;
        }
        __syncthreads();
    }
    if (active) {
        float4 mu[NS][2], rs[NS][2];
        if (NS == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mu[0][h] = ldcol4(b.mean, (cp + h * a.CP) * 4, a.H, 0.f);
                rs[0][h] = ldcol4(b.rstd, (cp + h * a.CP) * 4, a.H, 0.f);
            }
        }
        int cc[2];
        chunk_cols(a, cp, cc);
        for (int64_t row = r0 + rsub; row < r1; row += rstep) {
            float4 yv[2];
            yv[0] = ld4<VEC>(a.y, a.ldy, row, cc[0], a.H); yv[1] = ld4<VEC>(a.y, a.ldy, row, cc[1], a.H);
            int opaque = 0;
            if (NS == 2) asm volatile("" : "+v"(opaque));        // keeps the constant fetches below inside the loop
#pragma unroll
            for (int set = 0; set < NS; ++set) {
                float4 gv[2], d[2];
                gv[0] = ldg4<VEC>(b, row, cc[0], a.H, set); gv[1] = ldg4<VEC>(b, row, cc[1], a.H, set);
                if (NS == 2 && set == 1 && a.y2) {       // activated inputs: every set has a matrix of its own
                    yv[0] = ld4<VEC>(a.y2, a.ldy, row, cc[0], a.H); yv[1] = ld4<VEC>(a.y2, a.ldy, row, cc[1], a.H);
                }
                apply_d2(a, row, cp, yv, d, nullptr, set);
                if (NS == 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float* base = cst + opaque + set * 2 * (a.CH * 4) + cc[h];
                        mu[set][h] = *reinterpret_cast<const float4*>(base);
                        rs[set][h] = *reinterpret_cast<const float4*>(base + a.CH * 4);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 g = gv[h];
                    S1[set][h].x += g.x; S1[set][h].y += g.y; S1[set][h].z += g.z; S1[set][h].w += g.w;
                    S2[set][h].x += g.x * ((d[h].x - mu[set][h].x) * rs[set][h].x); S2[set][h].y += g.y * ((d[h].y - mu[set][h].y) * rs[set][h].y);
                    S2[set][h].z += g.z * ((d[h].z - mu[set][h].z) * rs[set][h].z); S2[set][h].w += g.w * ((d[h].w - mu[set][h].w) * rs[set][h].w);
                }
            }
        }
    }
    const int Hp = a.CH * 4;
#pragma unroll
    for (int set = 0; set < NS; ++set) {
        if (set > 0 || NS == 2) __syncthreads();        // NS = 2: every thread is done with the constants held in sh2
        sh[2 * t] = S1[set][0]; sh[2 * t + 1] = S1[set][1]; sh2[2 * t] = S2[set][0]; sh2[2 * t + 1] = S2[set][1];
        __syncthreads();
        if (active && rsub == 0) {
            double D1[2][4], D2[2][4];       // row slots added in fp64 (up to 256 of them for narrow rows)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                D1[h][0] = S1[set][h].x; D1[h][1] = S1[set][h].y; D1[h][2] = S1[set][h].z; D1[h][3] = S1[set][h].w;
                D2[h][0] = S2[set][h].x; D2[h][1] = S2[set][h].y; D2[h][2] = S2[set][h].z; D2[h][3] = S2[set][h].w;
            }
            for (int r = 1; r < a.RPB; ++r) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 u = sh[2 * (r * a.CP + cp) + h], w = sh2[2 * (r * a.CP + cp) + h];
                    D1[h][0] += u.x; D1[h][1] += u.y; D1[h][2] += u.z; D1[h][3] += u.w;
                    D2[h][0] += w.x; D2[h][1] += w.y; D2[h][2] += w.z; D2[h][3] += w.w;
                }
            }
            float* dst = partial + (int64_t)set * gridDim.x * 2 * Hp;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) break;
                const int c0 = (cp + h * a.CP) * 4;
                *reinterpret_cast<float4*>(dst + ((int64_t)blockIdx.x * 2 + 0) * Hp + c0) =
                    make_float4((float)D1[h][0], (float)D1[h][1], (float)D1[h][2], (float)D1[h][3]);
                *reinterpret_cast<float4*>(dst + ((int64_t)blockIdx.x * 2 + 1) * Hp + c0) =
                    make_float4((float)D2[h][0], (float)D2[h][1], (float)D2[h][2], (float)D2[h][3]);
            }
        }
    }
}

// sums of set s at sums + s * 2 * Hp; dgamma / dbeta take set 0 as `accumulate` says and every further set added
__global__ __launch_bounds__(1024) void k_ep_bwd_finalize(const float* __restrict__ partial, int nblocks, int H,
                                                         int Hp, float* __restrict__ sums,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         int accumulate, int nsets) {
    __shared__ double sh[FIN_SL][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    for (int set = 0; set < nsets; ++set) {
        const float* part = partial + (int64_t)set * nblocks * 2 * Hp;
        double a1 = 0.0, a2 = 0.0;
        if (c < H) {
            for (int b0 = sl; b0 < nblocks; b0 += 8 * FIN_SL) {
                float m[8], q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = min(b0 + u * FIN_SL, nblocks - 1);
                    m[u] = part[((int64_t)b * 2 + 0) * Hp + c];
                    q[u] = part[((int64_t)b * 2 + 1) * Hp + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = b0 + u * FIN_SL < nblocks;
                    a1 += in ? (double)m[u] : 0.0;
                    a2 += in ? (double)q[u] : 0.0;
                }
            }
        }
        const double s1 = fin_tree(sh, sl, cl, a1);
        const double s2 = fin_tree(sh, sl, cl, a2);
        if (sl == 0 && c < Hp) {
            sums[set * 2 * Hp + c] = (float)s1;
            sums[set * 2 * Hp + Hp + c] = (float)s2;
            if (c < H) {
                const bool add = accumulate || set > 0;
                if (dbeta) dbeta[c] = add ? dbeta[c] + (float)s1 : (float)s1;
                if (dgamma) dgamma[c] = add ? dgamma[c] + (float)s2 : (float)s2;
            }
        }
    }
}

// NS = 2 keeps the per-column constants of both sets (mean, rstd, sum g, sum g*xhat) in LDS and fetches them inside
// the row loop (launch with 2 * 4 * Hp floats of dynamic LDS; none for NS = 1).  With up to seven matrices in flight per
// row it still needs more than 128 VGPRs (forced into 128 it spilled 30 of them: 2.7 x its output in scratch writes), so it
// runs three workgroups per CU, on a grid of EP_WGS_TWO_SETS so that one round holds every workgroup.
template <bool VEC, int NS>
__global__ __launch_bounds__(256, NS == 2 ? 3 : 4) void k_ep_bwd_apply(const EpArgs a, const BwdArgs b) {
    extern __shared__ __attribute__((aligned(16))) float cst[];
    const int t = threadIdx.x;
    const int Hp = a.CH * 4;
    if (NS == 2) {
        for (int i = t; i < 2 * 4 * Hp; i += 256) {
            const int set = i / (4 * Hp), arr = (i / Hp) & 3, c = i % Hp;
            cst[i] = arr == 0 ? (c < a.H ? b.mean[set * a.H + c] : 0.f) : arr == 1 ? (c < a.H ? b.rstd[set * a.H + c] : 0.f)
                   : b.sums[set * 2 * Hp + (arr - 2) * Hp + c];
        }
        __syncthreads();
    }
    if (t >= a.RPB * a.CP) return;
    const int rsub = t / a.CP, cp = t - rsub * a.CP;
    const int nh = cp + a.CP < a.CH ? 2 : 1;
    const int64_t r0 = (int64_t)blockIdx.x * a.RPB, r1 = a.N, rstep = (int64_t)gridDim.x * a.RPB;       // see ep_block_rows
    const float invn = 1.f / (float)a.N;
    float4 mu[NS][2], rs[NS][2], ga[2], sg[NS][2], sgx[NS][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (cp + h * a.CP) * 4;
        ga[h] = ldcol4(b.gamma, c0, a.H, 1.f);
        if (NS == 1) {
            mu[0][h] = ldcol4(b.mean, c0, a.H, 0.f); rs[0][h] = ldcol4(b.rstd, c0, a.H, 0.f);
            sg[0][h] = sgx[0][h] = make_float4(0, 0, 0, 0);
            if (h < nh) {
                sg[0][h] = *reinterpret_cast<const float4*>(b.sums + c0);
                sgx[0][h] = *reinterpret_cast<const float4*>(b.sums + Hp + c0);
            }
        }
    }
    int cc[2];
    chunk_cols(a, cp, cc);
    for (int64_t row = r0 + rsub; row < r1; row += rstep) {
        float4 y[2], o[2];
        y[0] = ld4<VEC>(a.y, a.ldy, row, cc[0], a.H); y[1] = ld4<VEC>(a.y, a.ldy, row, cc[1], a.H);
        int opaque = 0;
        if (NS == 2) asm volatile("" : "+v"(opaque));            // keeps the constant fetches below inside the loop
#pragma unroll
        for (int set = 0; set < NS; ++set) {
            float4 d[2], k[2], gv[2];
            gv[0] = ldg4<VEC>(b, row, cc[0], a.H, set); gv[1] = ldg4<VEC>(b, row, cc[1], a.H, set);
            if (NS == 2 && set == 1 && a.y2) {           // activated inputs: every set has a matrix of its own
                y[0] = ld4<VEC>(a.y2, a.ldy, row, cc[0], a.H); y[1] = ld4<VEC>(a.y2, a.ldy, row, cc[1], a.H);
            }
            apply_d2(a, row, cp, y, d, k, set);
            if (NS == 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float* base = cst + opaque + set * 4 * Hp + cc[h];
                    mu[set][h] = *reinterpret_cast<const float4*>(base);
                    rs[set][h] = *reinterpret_cast<const float4*>(base + Hp);
                    sg[set][h] = *reinterpret_cast<const float4*>(base + 2 * Hp);
                    sgx[set][h] = *reinterpret_cast<const float4*>(base + 3 * Hp);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 g = gv[h];
#define HGV_EP_BWD(f)                                                                              \
                {                                                                                  \
                    const float xh = (d[h].f - mu[set][h].f) * rs[set][h].f;                       \
                    float dd = ga[h].f * rs[set][h].f * (g.f - sg[set][h].f * invn - xh * (sgx[set][h].f * invn)); \
                    dd *= k[h].f;                                                                  \
                    if (a.relu && !(y[h].f > 0.f)) dd = 0.f;                                       \
                    o[h].f = set ? o[h].f + dd : dd;                                               \
                }
                HGV_EP_BWD(x) HGV_EP_BWD(y) HGV_EP_BWD(z) HGV_EP_BWD(w)
#undef HGV_EP_BWD
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h < nh) st4<VEC>(b.dy, b.lddy, row, cc[h], a.H, o[h]);
    }
}

bool vec4_ok(const float* p, int64_t ld, int Hp) { return p == nullptr || (hgv_aligned(p, 16) && ld % 4 == 0 && ld >= Hp); }

int make_args(EpArgs& a, const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
              uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int* nblocks) {
    HGV_REQUIRE(N > 0 && H > 0 && H <= 2048, HGV_EINVAL, "bn_act: need N > 0 and 0 < H <= 2048 (N=%lld H=%d)", (long long)N, H);
    HGV_REQUIRE(y && ldy >= H, HGV_EINVAL, "bn_act: bad input buffer");
    HGV_REQUIRE(p >= 0.f && p < 1.f, HGV_EINVAL, "bn_act: dropout p=%f outside [0,1)", p);
    a.y = y; a.ldy = ldy; a.N = N; a.H = H; a.CH = (H + 3) / 4; a.CP = (a.CH + 1) / 2; a.RPB = 256 / a.CP;
    a.relu = relu; a.p = p; a.scale = 1.0f / (float)(1.0 - (double)p);
    a.thresh = (uint32_t)((double)p * 65536.0 + 0.5);
    a.mask = mask; a.seed = seed; a.offset = offset; a.offset2 = 0; a.offset_dev = offset_dev;
    a.mode = (p > 0.f) ? (mask ? 1 : 2) : 0;
    a.pre = 0; a.y2 = nullptr;
    // enough workgroups to fill the chip (8 per CU), at least 4 rows (of two chunks) per thread - two rows per thread for
    // small matrices, where the launch is as long as its longest workgroup (4 416 x 138: 4 % of a captured step against
    // 4 rows; one row per thread is slower again at width 256); which rows a workgroup takes: ep_block_rows
    int64_t nb = hgv_cdiv(N, (int64_t)a.RPB * (N >= 32768 ? 4 : 2));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    *nblocks = (int)nb;
    return HGV_OK;
}

}  // namespace

extern "C" size_t hgv_bn_act_ws(int64_t N, int H) {
    (void)N;
    const size_t Hp = (size_t)((H + 3) / 4) * 4;
    return hgv_align256(2048 * 2 * Hp * sizeof(float)) + hgv_align256(4 * Hp * sizeof(float));      // partials of two sets of <= 1024 row ranges; sums of two sets
}

namespace {
// Launch sequence of the training-mode forward; partial = [nsets][nblocks <= 1024][2][Hp] floats of scratch.  With
// nsets = 2 (a.offset / a.offset2, Philox noise, 16-byte rows) out2 receives the second set and save_mean / save_rstd
// hold [2][H].
int bn_act_fwd_launch(const EpArgs& a, int nb, const float* gamma, const float* beta, float eps, float* running_mean,
                      float* running_var, float momentum, int64_t* num_batches_tracked, float* save_mean, float* save_rstd,
                      float* out, int64_t ldo, float* partial, hipStream_t stream, int nsets = 1, float* out2 = nullptr) {
    const int Hp = a.CH * 4;
    const bool vec = vec4_ok(a.y, a.ldy, Hp) && vec4_ok(out, ldo, Hp) && vec4_ok(out2, ldo, Hp);
    HGV_REQUIRE(nsets == 1 || vec, HGV_EUNSUPPORTED, "bn_act_fwd2: rows must be 16-byte aligned with padded leading dimensions");
    if (nsets == 2) hipLaunchKernelGGL((k_ep_stats<true, 2>), dim3(nb), dim3(256), 0, stream, a, partial);
    else if (vec) hipLaunchKernelGGL((k_ep_stats<true, 1>), dim3(nb), dim3(256), 0, stream, a, partial);
    else hipLaunchKernelGGL((k_ep_stats<false, 1>), dim3(nb), dim3(256), 0, stream, a, partial);
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((a.H + 15) / 16), dim3(1024), 0, stream, partial, nb, a.RPB, a.N,
                       a.H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd, num_batches_tracked, nsets,
                       nullptr, nullptr, nullptr, nullptr);
    NormArgs n{save_mean, save_rstd, 0, eps, gamma, beta, out, ldo, out2};
    if (nsets == 2) hipLaunchKernelGGL((k_ep_apply<true, 2>), dim3(nb), dim3(256), 0, stream, a, n);
    else if (vec) hipLaunchKernelGGL((k_ep_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    else hipLaunchKernelGGL((k_ep_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    HGV_CHECK_LAUNCH("bn_act_fwd");
    return HGV_OK;
}

}  // namespace

// Training-mode forward.  num_batches_tracked (nullable device int64) is incremented.  mask: explicit keep mask [N,H] (uint8, contiguous) or null; with
// p > 0 and no mask the keep decisions come from Philox4x32 keyed by (seed, offset, row, chunk pair),
// 16 random bits per element, and the backward pass regenerates them from the same (seed, offset).
extern "C" int hgv_bn_act_fwd(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
                              uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma,
                              const float* beta, float eps,
                              float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                              float* save_mean, float* save_rstd, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(out && ldo >= H && save_mean && save_rstd && ws, HGV_EINVAL, "bn_act_fwd: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_fwd: workspace too small or misaligned");
    return bn_act_fwd_launch(a, nb, gamma, beta, eps, running_mean, running_var, momentum, num_batches_tracked, save_mean,
                             save_rstd, out, ldo, (float*)ws, stream);
}

// ---- the same forward for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the partial column statistics live in a device array of this library instead of a caller workspace.
// Every GPU the library runs on holds its own copy of the array (2.25 MiB) until the process ends.
#define HGV_BN_STATIC_MAX_H 288
#define HGV_BN_STATIC_MAX_BLOCKS 1024      // make_args never uses more row ranges than this
static __device__ __align__(256) float scratch[(size_t)HGV_BN_STATIC_MAX_BLOCKS * 2 * HGV_BN_STATIC_MAX_H];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of this GPU's copy of `scratch`, resolved once per device with that device current.  The array's
// name is only a host-side handle: kernels must be given this address, never the name.
int scratch_of_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    HGV_REQUIRE(e == hipSuccess && dev >= 0 && dev < kMaxDevices, HGV_ELAUNCH, "bn_act_fwd_static: hipGetDevice failed (%s)",
                hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        HGV_REQUIRE(e == hipSuccess && p != nullptr, HGV_ELAUNCH, "bn_act_fwd_static: hipGetSymbolAddress failed on device %d (%s)",
                    dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return HGV_OK;
}
}  // namespace

extern "C" int hgv_bn_act_fwd_static(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
                                     uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma,
                                     const float* beta, float eps, float* running_mean, float* running_var,
                                     float momentum, int64_t* num_batches_tracked, float* save_mean, float* save_rstd,
                                     float* out, int64_t ldo, hipStream_t stream) {
    HGV_REQUIRE(H <= HGV_BN_STATIC_MAX_H, HGV_EUNSUPPORTED,
                "bn_act_fwd_static: H=%d exceeds the built-in scratch (H <= %d); use hgv_bn_act_fwd with a workspace", H,
                HGV_BN_STATIC_MAX_H);
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(out && ldo >= H && save_mean && save_rstd, HGV_EINVAL, "bn_act_fwd_static: null/short buffer");
    HGV_REQUIRE(nb <= HGV_BN_STATIC_MAX_BLOCKS, HGV_EUNSUPPORTED, "bn_act_fwd_static: %d row ranges exceed the built-in scratch", nb);
    float* partial = nullptr;
    rc = scratch_of_current_device(&partial);
    if (rc != HGV_OK) return rc;
    // every row range [0, nb) writes its (mean, M2) pair before the merge kernel reads it: nothing has to start at zero
    return bn_act_fwd_launch(a, nb, gamma, beta, eps, running_mean, running_var, momentum, num_batches_tracked, save_mean,
                             save_rstd, out, ldo, partial, stream);
}

// Eval-mode forward: no dropout, running statistics.
extern "C" int hgv_bn_act_eval(const float* y, int64_t ldy, int64_t N, int H, int relu, const float* gamma,
                               const float* beta, float eps, const float* running_mean, const float* running_var,
                               float* out, int64_t ldo, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, 0.f, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(out && ldo >= H && running_mean && running_var, HGV_EINVAL, "bn_act_eval: null/short buffer");
    const int Hp = a.CH * 4;
    NormArgs n{running_mean, running_var, 1, eps, gamma, beta, out, ldo, nullptr};
    if (vec4_ok(y, ldy, Hp) && vec4_ok(out, ldo, Hp)) hipLaunchKernelGGL((k_ep_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    else hipLaunchKernelGGL((k_ep_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    HGV_CHECK_LAUNCH("bn_act_eval");
    return HGV_OK;
}

// Backward of the training-mode forward: dy (w.r.t. the pre-activation), dgamma, dbeta
// (accumulate != 0: added to dgamma / dbeta, i.e. straight into existing .grad buffers).  g2 (nullable): a second
// incoming gradient of the same shape, summed with g on the fly (an output with two consumers).
extern "C" int hgv_bn_act_bwd(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* y, int64_t ldy, int64_t N, int H, int relu,
                              float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                              const uint64_t* offset_dev, const float* gamma,
                              const float* save_mean, const float* save_rstd, float* dy, int64_t lddy,
                              float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(g && ldg >= H && dy && lddy >= H && save_mean && save_rstd && ws && (!g2 || ldg2 >= H), HGV_EINVAL, "bn_act_bwd: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd: workspace too small or misaligned");
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    BwdArgs b{g, ldg, g2, ldg2, save_mean, save_rstd, gamma, sums, dy, lddy, nullptr, 0, nullptr, 0};
    const bool vec = vec4_ok(y, ldy, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(g2, ldg2, Hp) && vec4_ok(dy, lddy, Hp);
    if (vec) hipLaunchKernelGGL((k_ep_bwd_stats<true, 1>), dim3(nb), dim3(256), 0, stream, a, b, partial);
    else hipLaunchKernelGGL((k_ep_bwd_stats<false, 1>), dim3(nb), dim3(256), 0, stream, a, b, partial);
    hipLaunchKernelGGL(k_ep_bwd_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, H, Hp, sums, dgamma, dbeta, accumulate, 1);
    if (vec) hipLaunchKernelGGL((k_ep_bwd_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    else hipLaunchKernelGGL((k_ep_bwd_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd");
    return HGV_OK;
}

#define HGV_BN_PAIR_MAX_H 512      /* the two-draw backward kernels keep 4 * Hp column constants in an 8 KB LDS array */

// ---- levels whose normalisation is left to their consumers ----------------------------------------------------------------
// The pass that takes a level's column statistics can leave d = dropout(relu(y)) behind in place of y.  Then the
// normalisation out = d * col_s + col_t (col_s = gamma * rstd, col_t = beta - mean * col_s) is handed to the consumers as
// two vectors - a following projection folds them into its weights (hgv_pack16h_weights_affine), a mean aggregation
// applies them to its finished rows (hgv_agg_fwd_affine) - and `out` is never written
// (gvae/gvae_skipcon.py:105-107 -> :110, :124-130: Dropout / BatchNorm1d outputs consumed by the next SAGEConv / Linear).
// hgv_bn_stats_act: y [N, H] is overwritten with d = dropout(relu(y)) (relu / p / mask / Philox arguments as for
// hgv_bn_act_fwd; without a relu a dropped element is stored as -0.0f and a kept zero as +0.0f) in the same pass that takes
// the column statistics of d: save_mean / save_rstd, the running-buffer update and col_s / col_t (4 * ceil(H / 4) floats
// each, pad columns zero).  One read and one write of the matrix instead of the two reads and one write of
// hgv_bn_act_fwd, and nothing of the level is written again.
extern "C" int hgv_bn_stats_act(float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask, uint64_t seed,
                                uint64_t offset, const uint64_t* offset_dev, const float* gamma, const float* beta, float eps,
                                float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                                float* save_mean, float* save_rstd, float* col_s, float* col_t, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(save_mean && save_rstd && col_s && col_t && ws, HGV_EINVAL, "bn_stats_act: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_stats_act: workspace too small or misaligned");
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    if (vec4_ok(y, ldy, Hp)) hipLaunchKernelGGL((k_ep_stats<true, 1, true>), dim3(nb), dim3(256), 0, stream, a, partial, y);
    else hipLaunchKernelGGL((k_ep_stats<false, 1, true>), dim3(nb), dim3(256), 0, stream, a, partial, y);
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, a.RPB, a.N,
                       a.H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd, num_batches_tracked, 1,
                       gamma, beta, col_s, col_t);
    HGV_CHECK_LAUNCH("bn_stats_act");
    return HGV_OK;
}

namespace {
// Keep decisions of one dropout call as bits: bit c & 31 of bits[row * ldb + (c >> 5)] = 1 when element (row, c) is kept
// (the decisions of keep8 with the same seed / offset; columns >= H read as kept).  The thread mapping is that of the
// other kernels of this file (a thread owns the chunk pair cp, cp + CP of a row); the row's words are assembled in LDS.
__device__ __forceinline__ uint32_t byte_nibble(uint32_t w, uint32_t T8) {
    return ((w & 0xffu) >= T8 ? 1u : 0u) | (((w >> 8) & 0xffu) >= T8 ? 2u : 0u) | (((w >> 16) & 0xffu) >= T8 ? 4u : 0u) | ((w >> 24) >= T8 ? 8u : 0u);
}

// BYTES: the rate is a multiple of 1 / 256 and the noise comes from the generator: a thread owns TWO chunk pairs (2 q, 2 q + 1)
// of a row and makes the one call that decides both (keep_bytes16); launched with RPB2 = 256 / ceil(CP / 2) rows per round.
template <bool BYTES>
__global__ __launch_bounds__(256) void k_dropout_bits(const EpArgs a, uint32_t* __restrict__ bits, int64_t ldb, int W, int RPB2) {
    __shared__ uint32_t sh[1024];
    const int t = threadIdx.x;
    const int per_row = BYTES ? (a.CP + 1) / 2 : a.CP, rpb = BYTES ? RPB2 : a.RPB;
    const bool active = t < rpb * per_row;
    const int rsub = t / per_row, cp = t - rsub * per_row;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < a.N; r0 += (int64_t)gridDim.x * rpb) {
        for (int e = t; e < rpb * W; e += 256) sh[e] = 0u;
        __syncthreads();
        const int64_t row = r0 + rsub;
        if (active && row < a.N) {
            uint32_t nibs[4];
            int chunk[4];
            int n = 2;
            if (BYTES) {
                const HgvPhilox ph(a.seed);
                const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
                const uint4 r = keep_bytes16(a, ph, row, cp, off);
                const uint32_t T8 = a.thresh >> 8;
                nibs[0] = byte_nibble(r.x, T8); nibs[1] = byte_nibble(r.y, T8); nibs[2] = byte_nibble(r.z, T8); nibs[3] = byte_nibble(r.w, T8);
                chunk[0] = 2 * cp; chunk[1] = 2 * cp + a.CP; chunk[2] = 2 * cp + 1; chunk[3] = 2 * cp + 1 + a.CP;
                if (2 * cp + 1 >= a.CP) chunk[2] = chunk[3] = a.CH;          // an odd pair count: the last thread owns one pair
                n = 4;
            } else {
                float4 k[2];
                keep8(a, row, cp, k);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    nibs[h] = (k[h].x != 0.f ? 1u : 0u) | (k[h].y != 0.f ? 2u : 0u) | (k[h].z != 0.f ? 4u : 0u) | (k[h].w != 0.f ? 8u : 0u);
                    chunk[h] = cp + h * a.CP;
                }
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (h >= n || chunk[h] >= a.CH) continue;
                const int c0 = chunk[h] * 4;
                uint32_t nib = nibs[h];
                if (c0 + 1 >= a.H) nib |= 2u;
                if (c0 + 2 >= a.H) nib |= 4u;
                if (c0 + 3 >= a.H) nib |= 8u;
                atomicOr(&sh[rsub * W + (c0 >> 5)], nib << (c0 & 31));      // distinct bits per thread: order is irrelevant
            }
        }
        __syncthreads();
        for (int e = t; e < rpb * W; e += 256) {
            const int64_t rr = r0 + e / W;
            if (rr < a.N) bits[rr * ldb + (e % W)] = sh[e];
        }
        __syncthreads();
    }
}
}  // namespace

// Keep bits of one dropout call over [N, H]: the decisions hgv_bn_stats_act / hgv_bn_act_fwd take for the same explicit
// mask ([N, H] bytes, non-zero = keep) or, with mask == nullptr, the same Philox (p, seed, offset, offset_dev).
// bits: [N, ld_bits] 32-bit words, ld_bits >= ceil(4 * ceil(H / 4) / 32); bits of columns
// in [H, 32 * ld_bits) covered by a chunk are set (kept).  For projections that store the activated level themselves.
extern "C" int hgv_dropout_bits(int64_t N, int H, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, uint32_t* bits, int64_t ld_bits, hipStream_t stream) {
    EpArgs a; int nb;
    HGV_REQUIRE(bits != nullptr && p > 0.f, HGV_EINVAL, "dropout_bits: null buffer or p = 0");
    int rc = make_args(a, reinterpret_cast<const float*>(bits), H, N, H, 0, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    const int W = (a.CH * 4 + 31) / 32;
    HGV_REQUIRE(ld_bits >= W, HGV_EINVAL, "dropout_bits: %lld words per row, %d needed", (long long)ld_bits, W);
    const bool bytes = a.mode == 2 && (a.thresh & 0xffu) == 0u;
    const int rpb = bytes ? 256 / ((a.CP + 1) / 2) : a.RPB;
    HGV_REQUIRE(rpb * W <= 1024, HGV_EUNSUPPORTED, "dropout_bits: H=%d too wide", H);
    int64_t g = hgv_cdiv(N, rpb);
    if (g > 2048) g = 2048;
    if (bytes) hipLaunchKernelGGL(k_dropout_bits<true>, dim3((unsigned)g), dim3(256), 0, stream, a, bits, ld_bits, W, rpb);
    else hipLaunchKernelGGL(k_dropout_bits<false>, dim3((unsigned)g), dim3(256), 0, stream, a, bits, ld_bits, W, rpb);
    HGV_CHECK_LAUNCH("dropout_bits");
    return HGV_OK;
}

// The statistics half of hgv_bn_stats_act for a matrix that already holds d = dropout(relu(y)) (a projection stored it:
// hgv_gemm_nt16h_act): one read of d, no write; same partial sums, hence the same mean / rstd / col_s / col_t / running
// statistics bit for bit.
extern "C" int hgv_bn_stats_pre(const float* d, int64_t ldd, int64_t N, int H, const float* gamma, const float* beta, float eps,
                                float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                                float* save_mean, float* save_rstd, float* col_s, float* col_t, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, d, ldd, N, H, 0, 0.f, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(save_mean && save_rstd && col_s && col_t && ws, HGV_EINVAL, "bn_stats_pre: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_stats_pre: workspace too small or misaligned");
    a.pre = 1;
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    if (vec4_ok(d, ldd, Hp)) hipLaunchKernelGGL((k_ep_stats<true, 1>), dim3(nb), dim3(256), 0, stream, a, partial);
    else hipLaunchKernelGGL((k_ep_stats<false, 1>), dim3(nb), dim3(256), 0, stream, a, partial);
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, a.RPB, a.N,
                       a.H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd, num_batches_tracked, 1,
                       gamma, beta, col_s, col_t);
    HGV_CHECK_LAUNCH("bn_stats_pre");
    return HGV_OK;
}

// The merge step of the level statistics on its own, for partial statistics another kernel left (hgv_gemm_nt16h_act_stats):
// partial[nparts][2][Hp] = (mean, M2) of part p over the rows of the group_rows-row groups p, p + nparts, ... of [0, N).
// Writes what hgv_bn_stats_act writes (mean / rstd, col_s / col_t, running statistics).
extern "C" int hgv_bn_stats_merge(const float* partial, int nparts, int group_rows, int64_t N, int H, const float* gamma,
                                  const float* beta, float eps, float* running_mean, float* running_var, float momentum,
                                  int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* col_s, float* col_t,
                                  hipStream_t stream) {
    HGV_REQUIRE(partial && nparts > 0 && group_rows > 0 && N > 1 && H > 0 && H <= 2048, HGV_EINVAL, "bn_stats_merge: bad arguments");
    HGV_REQUIRE(save_mean && save_rstd && col_s && col_t, HGV_EINVAL, "bn_stats_merge: null buffer");
    const int Hp = (H + 3) / 4 * 4;
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nparts, group_rows, N,
                       H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd, num_batches_tracked, 1,
                       gamma, beta, col_s, col_t);
    HGV_CHECK_LAUNCH("bn_stats_merge");
    return HGV_OK;
}

// out = (d - mean) * rstd * gamma + beta for a consumer that needs the normalised matrix after all (same arithmetic as
// the second pass of hgv_bn_act_fwd).
extern "C" int hgv_bn_apply(const float* d, int64_t ldd, int64_t N, int H, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_rstd, float* out, int64_t ldo, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, d, ldd, N, H, 0, 0.f, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(out && ldo >= H && save_mean && save_rstd, HGV_EINVAL, "bn_apply: null/short buffer");
    a.pre = 1;
    const int Hp = a.CH * 4;
    NormArgs n{save_mean, save_rstd, 0, 0.f, gamma, beta, out, ldo, nullptr};
    if (vec4_ok(d, ldd, Hp) && vec4_ok(out, ldo, Hp)) hipLaunchKernelGGL((k_ep_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    else hipLaunchKernelGGL((k_ep_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, n);
    HGV_CHECK_LAUNCH("bn_apply");
    return HGV_OK;
}

// Backward of such a level: hgv_bn_act_bwd with the activated matrix d in place of y (relu and dropout are read off the
// stored values); dy is the gradient with respect to y; p as in the forward.
extern "C" int hgv_bn_act_bwd_pre(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* d, int64_t ldd,
                                  int64_t N, int H, int relu, float p, const float* gamma, const float* save_mean,
                                  const float* save_rstd, float* dy, int64_t lddy, float* dgamma, float* dbeta, int accumulate,
                                  void* ws, size_t ws_bytes, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, d, ldd, N, H, relu, p, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(g && ldg >= H && dy && lddy >= H && save_mean && save_rstd && ws && (!g2 || ldg2 >= H), HGV_EINVAL, "bn_act_bwd_pre: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd_pre: workspace too small or misaligned");
    a.pre = 1; a.mode = 0;
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    BwdArgs b{g, ldg, g2, ldg2, save_mean, save_rstd, gamma, sums, dy, lddy, nullptr, 0, nullptr, 0};
    const bool vec = vec4_ok(d, ldd, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(g2, ldg2, Hp) && vec4_ok(dy, lddy, Hp);
    if (vec) hipLaunchKernelGGL((k_ep_bwd_stats<true, 1>), dim3(nb), dim3(256), 0, stream, a, b, partial);
    else hipLaunchKernelGGL((k_ep_bwd_stats<false, 1>), dim3(nb), dim3(256), 0, stream, a, b, partial);
    hipLaunchKernelGGL(k_ep_bwd_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, H, Hp, sums, dgamma, dbeta, accumulate, 1);
    if (vec) hipLaunchKernelGGL((k_ep_bwd_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    else hipLaunchKernelGGL((k_ep_bwd_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd_pre");
    return HGV_OK;
}

// The same backward without the statistics pass.  When every consumer of the level is a projection (or a mean
// aggregation feeding one) that received the level as (d, s, t), the incoming gradient is a sum of products
//   g = sum_parts G_p W_p      (G_p: the consumer's own output gradient, or its transposed aggregate; W_p [n_p, H]),
// and the two column sums the backward needs follow from matrices the consumers' weight-gradient products have
// already formed - raw_p = G_p^T d [n_p, H] and csum_p = column sums of G_p [n_p]:
//   sum_i g_ic      = sum_p sum_n W_p[n,c] csum_p[n],
//   sum_i g_ic d_ic = sum_p sum_n W_p[n,c] raw_p[n,c],     sum_i g_ic xhat_ic = rstd_c (sum_i g_ic d_ic - mean_c sum_i g_ic).
// A few hundred multiply-adds per column (float64) replace a pass over g, g2 and d.
namespace {
constexpr int ALG_MAX_PARTS = 4;
struct AlgPart { const float* W; int64_t ldw; const float* raw; int64_t ldraw; const float* csum; int64_t cs_stride; int n; };
struct AlgArgs {
    AlgPart part[ALG_MAX_PARTS]; int nparts; int H, Hp;
    const float* mean; const float* rstd;
    float* sums; float* dgamma; float* dbeta; int accumulate;
};
// 32 columns x 32 row groups per workgroup: a part of 138 rows is five rows per thread, all of them in flight at once (the
// kernel is a chain of dependent-latency loads, not arithmetic: with eight row groups and four rows in flight it took 12 us)
constexpr int ALG_GROUPS = 32;
__global__ __launch_bounds__(32 * ALG_GROUPS) void k_ep_bwd_sums_alg(const AlgArgs a) {
    __shared__ double sh1[ALG_GROUPS][33], sh2[ALG_GROUPS][33];
    const int cl = threadIdx.x & 31, ng = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < a.H) {
        for (int p = 0; p < a.nparts; ++p) {
            const AlgPart& q = a.part[p];
            for (int n0 = ng; n0 < q.n; n0 += 8 * ALG_GROUPS) {             // eight rows of W in flight per thread
                float w[8], r[8], s[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int n = min(n0 + ALG_GROUPS * u, q.n - 1);
                    w[u] = q.W[(int64_t)n * q.ldw + c]; r[u] = q.raw[(int64_t)n * q.ldraw + c]; s[u] = q.csum[(int64_t)n * q.cs_stride];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (n0 + ALG_GROUPS * u < q.n) { a1 += (double)w[u] * (double)s[u]; a2 += (double)w[u] * (double)r[u]; }
                }
            }
        }
    }
    sh1[ng][cl] = a1; sh2[ng][cl] = a2;
    __syncthreads();
    if (ng == 0 && c < a.Hp) {
        double s1 = 0.0, sd = 0.0;
#pragma unroll
        for (int u = 0; u < ALG_GROUPS; ++u) { s1 += sh1[u][cl]; sd += sh2[u][cl]; }
        float f1 = 0.f, f2 = 0.f;
        if (c < a.H) {
            f1 = (float)s1;
            f2 = (float)((double)a.rstd[c] * (sd - (double)a.mean[c] * s1));
            if (a.dbeta) a.dbeta[c] = a.accumulate ? a.dbeta[c] + f1 : f1;
            if (a.dgamma) a.dgamma[c] = a.accumulate ? a.dgamma[c] + f2 : f2;
        }
        a.sums[c] = f1; a.sums[a.Hp + c] = f2;
    }
}
}  // namespace

// hgv_bn_act_bwd_pre with the two column sums taken from the consumers' products (see above) instead of a pass over the
// matrices: nparts <= 4 parts (W_p [n_p, H] with leading dimension ldw_p, raw_p [n_p, H] with ldraw_p, csum_p with element
// stride cs_stride_p).  g (+ g2) must be exactly sum_p G_p W_p; dy, dgamma, dbeta as for hgv_bn_act_bwd_pre.
extern "C" int hgv_bn_act_bwd_pre_alg(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* d, int64_t ldd,
                                      int64_t N, int H, int relu, float p, const float* gamma, const float* save_mean,
                                      const float* save_rstd, int nparts, const float* const* W, const int64_t* ldw,
                                      const float* const* raw, const int64_t* ldraw, const float* const* csum,
                                      const int64_t* cs_stride, const int* n, float* dy, int64_t lddy, float* dgamma,
                                      float* dbeta, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, d, ldd, N, H, relu, p, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(g && ldg >= H && dy && lddy >= H && save_mean && save_rstd && ws && (!g2 || ldg2 >= H), HGV_EINVAL, "bn_act_bwd_pre_alg: null/short buffer");
    HGV_REQUIRE(nparts >= 1 && nparts <= ALG_MAX_PARTS && W && ldw && raw && ldraw && csum && cs_stride && n, HGV_EINVAL,
                "bn_act_bwd_pre_alg: 1..%d parts expected", ALG_MAX_PARTS);
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd_pre_alg: workspace too small or misaligned");
    a.pre = 1; a.mode = 0;
    const int Hp = a.CH * 4;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    AlgArgs q{};
    q.nparts = nparts; q.H = H; q.Hp = Hp; q.mean = save_mean; q.rstd = save_rstd; q.sums = sums; q.dgamma = dgamma; q.dbeta = dbeta;
    q.accumulate = accumulate;
    for (int k = 0; k < nparts; ++k) {
        HGV_REQUIRE(W[k] && raw[k] && csum[k] && n[k] > 0 && ldw[k] >= H && ldraw[k] >= H, HGV_EINVAL, "bn_act_bwd_pre_alg: bad part %d", k);
        q.part[k] = AlgPart{W[k], ldw[k], raw[k], ldraw[k], csum[k], cs_stride[k], n[k]};
    }
    hipLaunchKernelGGL(k_ep_bwd_sums_alg, dim3((Hp + 31) / 32), dim3(32 * ALG_GROUPS), 0, stream, q);
    BwdArgs b{g, ldg, g2, ldg2, save_mean, save_rstd, gamma, sums, dy, lddy, nullptr, 0, nullptr, 0};
    const bool vec = vec4_ok(d, ldd, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(g2, ldg2, Hp) && vec4_ok(dy, lddy, Hp);
    if (vec) hipLaunchKernelGGL((k_ep_bwd_apply<true, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    else hipLaunchKernelGGL((k_ep_bwd_apply<false, 1>), dim3(nb), dim3(256), 0, stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd_pre_alg");
    return HGV_OK;
}

// The two column sums of hgv_bn_act_bwd_pre_alg by themselves: sums[2][4 ceil(H / 4)] = (sum_i g_ic, sum_i g_ic xhat_ic) from
// the consumers' parts, with dgamma / dbeta (nullable) written or added to as there.  For a consumer whose grad-input product
// applies the level's backward to what it stores (hgv_gemm_nt16h_bwd).
extern "C" int hgv_bn_bwd_sums_alg(int H, const float* save_mean, const float* save_rstd, int nparts, const float* const* W,
                                   const int64_t* ldw, const float* const* raw, const int64_t* ldraw, const float* const* csum,
                                   const int64_t* cs_stride, const int* n, float* sums, float* dgamma, float* dbeta, int accumulate,
                                   hipStream_t stream) {
    HGV_REQUIRE(H > 0 && H <= 2048 && save_mean && save_rstd && sums, HGV_EINVAL, "bn_bwd_sums_alg: bad arguments");
    HGV_REQUIRE(nparts >= 1 && nparts <= ALG_MAX_PARTS && W && ldw && raw && ldraw && csum && cs_stride && n, HGV_EINVAL,
                "bn_bwd_sums_alg: 1..%d parts expected", ALG_MAX_PARTS);
    const int Hp = (H + 3) / 4 * 4;
    AlgArgs q{};
    q.nparts = nparts; q.H = H; q.Hp = Hp; q.mean = save_mean; q.rstd = save_rstd; q.sums = sums; q.dgamma = dgamma; q.dbeta = dbeta;
    q.accumulate = accumulate;
    for (int k = 0; k < nparts; ++k) {
        HGV_REQUIRE(W[k] && raw[k] && csum[k] && n[k] > 0 && ldw[k] >= H && ldraw[k] >= H, HGV_EINVAL, "bn_bwd_sums_alg: bad part %d", k);
        q.part[k] = AlgPart{W[k], ldw[k], raw[k], ldraw[k], csum[k], cs_stride[k], n[k]};
    }
    hipLaunchKernelGGL(k_ep_bwd_sums_alg, dim3((Hp + 31) / 32), dim3(32 * ALG_GROUPS), 0, stream, q);
    HGV_CHECK_LAUNCH("bn_bwd_sums_alg");
    return HGV_OK;
}

// ---- the same for two dropout draws of one pre-activation (the two encode passes of a train step over the shared first
// convolution, gvae/gvae_skipcon.py:166-171): y is overwritten with the first draw's d, d2 receives the second's; one read
// of y.  save_mean / save_rstd hold [2][H], col_s / col_t [2][4 * ceil(H / 4)]; the running statistics take the first
// draw's update, then the second's.  Philox noise only (p > 0), 16-byte rows.
extern "C" int hgv_bn_stats_act2(float* y, float* d2, int64_t ldy, int64_t N, int H, int relu, float p, uint64_t seed,
                                 uint64_t offset, uint64_t offset2, const uint64_t* offset_dev, const float* gamma,
                                 const float* beta, float eps, float* running_mean, float* running_var, float momentum,
                                 int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* col_s, float* col_t,
                                 void* ws, size_t ws_bytes, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, nullptr, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(p > 0.f, HGV_EINVAL, "bn_stats_act2: two draws need p > 0");
    HGV_REQUIRE(H <= HGV_BN_PAIR_MAX_H, HGV_EUNSUPPORTED, "bn_stats_act2: H=%d exceeds %d", H, HGV_BN_PAIR_MAX_H);
    HGV_REQUIRE(d2 && save_mean && save_rstd && col_s && col_t && ws, HGV_EINVAL, "bn_stats_act2: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_stats_act2: workspace too small or misaligned");
    const int Hp = a.CH * 4;
    HGV_REQUIRE(vec4_ok(y, ldy, Hp) && vec4_ok(d2, ldy, Hp), HGV_EUNSUPPORTED, "bn_stats_act2: rows must be 16-byte aligned with padded leading dimensions");
    a.offset2 = offset2;
    float* partial = (float*)ws;
    hipLaunchKernelGGL((k_ep_stats<true, 2, true>), dim3(nb), dim3(256), 0, stream, a, partial, y, d2);
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, a.RPB, a.N,
                       a.H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd, num_batches_tracked, 2,
                       gamma, beta, col_s, col_t);
    HGV_CHECK_LAUNCH("bn_stats_act2");
    return HGV_OK;
}

// Backward of the two draws in one pass: g / g2 are the incoming gradients of the first draw's output (g2 nullable), h / h2
// those of the second's, d1 / d2 the two activated matrices, dy the SUM of the two input gradients.  nparts0 + nparts1 parts
// (hgv_bn_act_bwd_pre_alg; the first nparts0 belong to the first draw): with both counts > 0 the column sums come from the
// consumers' products, with both 0 from a pass over the matrices.
extern "C" int hgv_bn_act_bwd2_pre(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* h, int64_t ldh,
                                   const float* h2, int64_t ldh2, const float* d1, const float* d2, int64_t ldd, int64_t N,
                                   int H, int relu, float p, const float* gamma, const float* save_mean,
                                   const float* save_rstd, int nparts0, int nparts1, const float* const* W,
                                   const int64_t* ldw, const float* const* raw, const int64_t* ldraw,
                                   const float* const* csum, const int64_t* cs_stride, const int* n, float* dy, int64_t lddy,
                                   float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                                   hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, d1, ldd, N, H, relu, p, nullptr, 0, 0, nullptr, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(H <= HGV_BN_PAIR_MAX_H, HGV_EUNSUPPORTED, "bn_act_bwd2_pre: H=%d exceeds %d", H, HGV_BN_PAIR_MAX_H);
    HGV_REQUIRE(g && ldg >= H && h && ldh >= H && d2 && dy && lddy >= H && save_mean && save_rstd && ws && (!g2 || ldg2 >= H) &&
                (!h2 || ldh2 >= H), HGV_EINVAL, "bn_act_bwd2_pre: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd2_pre: workspace too small or misaligned");
    HGV_REQUIRE((nparts0 > 0) == (nparts1 > 0) && nparts0 >= 0 && nparts0 <= ALG_MAX_PARTS && nparts1 <= ALG_MAX_PARTS, HGV_EINVAL,
                "bn_act_bwd2_pre: parts for both draws or for none (at most %d each)", ALG_MAX_PARTS);
    a.pre = 1; a.mode = 0; a.y2 = d2;
    const int Hp = a.CH * 4;
    HGV_REQUIRE(vec4_ok(d1, ldd, Hp) && vec4_ok(d2, ldd, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(g2, ldg2, Hp) && vec4_ok(h, ldh, Hp) &&
                vec4_ok(h2, ldh2, Hp) && vec4_ok(dy, lddy, Hp), HGV_EUNSUPPORTED,
                "bn_act_bwd2_pre: rows must be 16-byte aligned with padded leading dimensions");
    float* partial = (float*)ws;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    BwdArgs b{g, ldg, g2, ldg2, save_mean, save_rstd, gamma, sums, dy, lddy, h, ldh, h2, ldh2};
    if (nparts0 > 0) {
        HGV_REQUIRE(W && ldw && raw && ldraw && csum && cs_stride && n, HGV_EINVAL, "bn_act_bwd2_pre: null part arrays");
        int first = 0;
        for (int set = 0; set < 2; ++set) {
            const int cnt = set ? nparts1 : nparts0;
            AlgArgs q{};
            q.nparts = cnt; q.H = H; q.Hp = Hp; q.mean = save_mean + set * H; q.rstd = save_rstd + set * H;
            q.sums = sums + set * 2 * Hp; q.dgamma = dgamma; q.dbeta = dbeta; q.accumulate = (accumulate || set) ? 1 : 0;
            for (int k = 0; k < cnt; ++k) {
                const int j = first + k;
                HGV_REQUIRE(W[j] && raw[j] && csum[j] && n[j] > 0 && ldw[j] >= H && ldraw[j] >= H, HGV_EINVAL, "bn_act_bwd2_pre: bad part %d", j);
                q.part[k] = AlgPart{W[j], ldw[j], raw[j], ldraw[j], csum[j], cs_stride[j], n[j]};
            }
            hipLaunchKernelGGL(k_ep_bwd_sums_alg, dim3((Hp + 31) / 32), dim3(32 * ALG_GROUPS), 0, stream, q);
            first += cnt;
        }
    } else {
        hipLaunchKernelGGL((k_ep_bwd_stats<true, 2>), dim3(nb), dim3(256), 0, stream, a, b, partial);
        hipLaunchKernelGGL(k_ep_bwd_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, H, Hp, sums, dgamma, dbeta, accumulate, 2);
    }
    hipLaunchKernelGGL((k_ep_bwd_apply<true, 2>), dim3(nb < EP_WGS_TWO_SETS ? nb : EP_WGS_TWO_SETS), dim3(256),
                       (size_t)2 * 4 * Hp * sizeof(float), stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd2_pre");
    return HGV_OK;
}

// Weight-gradient finish for operands that were handed over un-normalised.  With x = a * s + t column by column,
//   G^T x = (G^T a) diag(s) + colsum(G) t^T:
// dst[n,k] (+)= raw[n,k] * s[k] + colsum[n * cs_stride] * t[k]   (s / t null: dst (+)= raw).  Up to three jobs per launch
// (the two halves of a K-concatenated weight and the bias vector as an N x 1 matrix).  colsum with a stride: for an
// operand [mean(d) | indicator] the sum that multiplies t is the raw product's indicator column.
namespace {
struct FinishJob { float* dst; int64_t lddst; const float* raw; int64_t ldraw; int N, K; const float* s; const float* t; const float* colsum; int64_t cs_stride; };
struct FinishArgs { FinishJob job[3]; int njobs; int accumulate; };
__global__ __launch_bounds__(256) void k_wgrad_finish(const FinishArgs f) {
    const FinishJob& j = f.job[blockIdx.y];
    const int64_t total = (int64_t)j.N * j.K;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int n = (int)(e / j.K), k = (int)(e - (int64_t)n * j.K);
        float v = j.raw[(int64_t)n * j.ldraw + k];
        if (j.s) v = __fmaf_rn(v, j.s[k], j.colsum[(int64_t)n * j.cs_stride] * j.t[k]);
        float* d = j.dst + (int64_t)n * j.lddst + k;
        *d = f.accumulate ? *d + v : v;
    }
}
}  // namespace

extern "C" int hgv_wgrad_finish(int njobs, float* const* dst, const int64_t* lddst, const float* const* raw, const int64_t* ldraw,
                                const int* N, const int* K, const float* const* s, const float* const* t,
                                const float* const* colsum, const int64_t* cs_stride, int accumulate, hipStream_t stream) {
    HGV_REQUIRE(njobs >= 1 && njobs <= 3 && dst && lddst && raw && ldraw && N && K && s && t && colsum && cs_stride, HGV_EINVAL,
                "wgrad_finish: 1..3 jobs and non-null arrays expected");
    FinishArgs f{};
    f.njobs = njobs; f.accumulate = accumulate;
    int64_t most = 0;
    for (int q = 0; q < njobs; ++q) {
        HGV_REQUIRE(dst[q] && raw[q] && N[q] > 0 && K[q] > 0 && lddst[q] >= K[q] && ldraw[q] >= K[q], HGV_EINVAL, "wgrad_finish: bad job %d", q);
        HGV_REQUIRE((s[q] == nullptr) == (t[q] == nullptr) && (!s[q] || colsum[q]), HGV_EINVAL, "wgrad_finish: job %d needs s, t and colsum together", q);
        f.job[q] = FinishJob{dst[q], lddst[q], raw[q], ldraw[q], N[q], K[q], s[q], t[q], colsum[q], cs_stride[q]};
        most = (int64_t)N[q] * K[q] > most ? (int64_t)N[q] * K[q] : most;
    }
    int64_t blocks = hgv_cdiv(most, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)blocks, (unsigned)njobs), dim3(256), 0, stream, f);
    HGV_CHECK_LAUNCH("wgrad_finish");
    return HGV_OK;
}

// ---- two dropout draws of one pre-activation in one pass ------------------------------------------------------------
// The reference encodes every batch twice per train step (gvae/gvae_skipcon.py:166-171); nothing stochastic precedes the
// first convolution, so both passes apply relu -> Dropout -> BatchNorm1d (:105-107) to the SAME matrix y with different
// masks.  These entry points do the work of two hgv_bn_act_fwd / hgv_bn_act_bwd calls over one read of y: outputs out
// (Philox offset) and out2 (offset2), statistics save_mean / save_rstd = [2][H], the running statistics and
// num_batches_tracked updated for the first draw and then for the second; backward: g / g2 are the incoming gradients
// of out (g2 nullable), h / h2 those of out2 (h2 nullable), dy the SUM of the two input gradients, dgamma / dbeta the
// sums over both draws.  Results are those of the two separate calls bit for bit.  Philox noise with p > 0 only
// (explicit masks: two calls); buffers 16-byte aligned with leading dimensions that are multiples of 4.
extern "C" int hgv_bn_act_fwd2(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, uint64_t seed,
                               uint64_t offset, uint64_t offset2, const uint64_t* offset_dev, const float* gamma,
                               const float* beta, float eps, float* running_mean, float* running_var, float momentum,
                               int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* out, float* out2,
                               int64_t ldo, void* ws, size_t ws_bytes, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, nullptr, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(p > 0.f, HGV_EINVAL, "bn_act_fwd2: two draws need p > 0");
    HGV_REQUIRE(H <= HGV_BN_PAIR_MAX_H, HGV_EUNSUPPORTED, "bn_act_fwd2: H=%d exceeds %d (use two hgv_bn_act_fwd calls)", H, HGV_BN_PAIR_MAX_H);
    HGV_REQUIRE(out && out2 && ldo >= H && save_mean && save_rstd && ws, HGV_EINVAL, "bn_act_fwd2: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_fwd2: workspace too small or misaligned");
    a.offset2 = offset2;
    return bn_act_fwd_launch(a, nb, gamma, beta, eps, running_mean, running_var, momentum, num_batches_tracked, save_mean,
                             save_rstd, out, ldo, (float*)ws, stream, 2, out2);
}

extern "C" int hgv_bn_act_bwd2(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* h, int64_t ldh,
                               const float* h2, int64_t ldh2, const float* y, int64_t ldy, int64_t N, int H, int relu,
                               float p, uint64_t seed, uint64_t offset, uint64_t offset2, const uint64_t* offset_dev,
                               const float* gamma, const float* save_mean, const float* save_rstd, float* dy,
                               int64_t lddy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                               hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, nullptr, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(p > 0.f, HGV_EINVAL, "bn_act_bwd2: two draws need p > 0");
    HGV_REQUIRE(H <= HGV_BN_PAIR_MAX_H, HGV_EUNSUPPORTED, "bn_act_bwd2: H=%d exceeds %d (use two hgv_bn_act_bwd calls)", H, HGV_BN_PAIR_MAX_H);
    HGV_REQUIRE(g && ldg >= H && h && ldh >= H && dy && lddy >= H && save_mean && save_rstd && ws && (!g2 || ldg2 >= H) &&
                (!h2 || ldh2 >= H), HGV_EINVAL, "bn_act_bwd2: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd2: workspace too small or misaligned");
    a.offset2 = offset2;
    const int Hp = a.CH * 4;
    HGV_REQUIRE(vec4_ok(y, ldy, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(g2, ldg2, Hp) && vec4_ok(h, ldh, Hp) && vec4_ok(h2, ldh2, Hp) &&
                vec4_ok(dy, lddy, Hp), HGV_EUNSUPPORTED, "bn_act_bwd2: rows must be 16-byte aligned with padded leading dimensions");
    float* partial = (float*)ws;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    BwdArgs b{g, ldg, g2, ldg2, save_mean, save_rstd, gamma, sums, dy, lddy, h, ldh, h2, ldh2};
    hipLaunchKernelGGL((k_ep_bwd_stats<true, 2>), dim3(nb), dim3(256), 0, stream, a, b, partial);
    hipLaunchKernelGGL(k_ep_bwd_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, H, Hp, sums, dgamma, dbeta, accumulate, 2);
    hipLaunchKernelGGL((k_ep_bwd_apply<true, 2>), dim3(nb < EP_WGS_TWO_SETS ? nb : EP_WGS_TWO_SETS), dim3(256),
                       (size_t)2 * 4 * Hp * sizeof(float), stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd2");
    return HGV_OK;
}
