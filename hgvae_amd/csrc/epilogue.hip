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
// Column statistics: each workgroup reduces a row range with sums shifted by the range's
// first row (no catastrophic cancellation), writes (mean_b, M2_b); a second tiny kernel
// merges the partials with Chan's formula in fp64.  No atomics -> bitwise reproducible.
//
// Thread mapping: a thread owns one 4-column chunk; 256/CH row slots run side by side so
// consecutive threads read consecutive 16-byte pieces of consecutive rows.
#include "common.h"

namespace {

struct EpArgs {
    const float* y; int64_t ldy;
    int64_t N; int H; int CH; int RPB; int64_t rows_per_block;
    int relu; int mode;              // mode 0: no dropout, 1: explicit mask, 2: Philox
    float p; float scale;
    const uint8_t* mask;             // [N, H] contiguous
    uint64_t seed, offset;
    const uint64_t* offset_dev;      // optional device word added to offset (advanced once per step under graph replay)
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

// keep-multipliers (0 or scale) for the 4 columns of a chunk
__device__ __forceinline__ float4 keep4(const EpArgs& a, int64_t row, int chunk) {
    float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
    const int c0 = chunk * 4;
    if (a.mode == 1) {
        const uint8_t* m = a.mask + row * a.H + c0;
        k.x = (c0 + 0 < a.H && m[0]) ? a.scale : 0.f;
        k.y = (c0 + 1 < a.H && m[1]) ? a.scale : 0.f;
        k.z = (c0 + 2 < a.H && m[2]) ? a.scale : 0.f;
        k.w = (c0 + 3 < a.H && m[3]) ? a.scale : 0.f;
    } else if (a.mode == 2) {
        const HgvPhilox ph(a.seed);
        const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
        const uint4 r = ph((uint32_t)row, (uint32_t)((uint64_t)row >> 32) ^ ((uint32_t)chunk << 8),
                           (uint32_t)off, (uint32_t)(off >> 32));
        k.x = hgv_u01(r.x) >= a.p ? a.scale : 0.f;
        k.y = hgv_u01(r.y) >= a.p ? a.scale : 0.f;
        k.z = hgv_u01(r.z) >= a.p ? a.scale : 0.f;
        k.w = hgv_u01(r.w) >= a.p ? a.scale : 0.f;
    }
    return k;
}

template <bool VEC>
__device__ __forceinline__ float4 load_d(const EpArgs& a, int64_t row, int chunk, float4* y_out = nullptr,
                                         float4* keep_out = nullptr) {
    float4 v = ld4<VEC>(a.y, a.ldy, row, chunk * 4, a.H);
    if (y_out) *y_out = v;
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (a.mode != 0) {
        const float4 k = keep4(a, row, chunk);
        if (keep_out) *keep_out = k;
        v.x *= k.x; v.y *= k.y; v.z *= k.z; v.w *= k.w;
    } else if (keep_out) {
        *keep_out = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    return v;
}

// partial layout: [block][2][Hp] = (mean_b, M2_b), Hp = CH*4
template <bool VEC>
__global__ __launch_bounds__(256) void k_ep_stats(const EpArgs a, float* __restrict__ partial) {
    __shared__ float4 sh[256];
    __shared__ float4 sh2[256];
    const int t = threadIdx.x;
    const bool active = t < a.RPB * a.CH;
    const int rsub = t / a.CH, chunk = t - rsub * a.CH;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = min(a.N, r0 + a.rows_per_block);
    if (active && rsub == 0) sh[chunk] = load_d<VEC>(a, r0, chunk);
    __syncthreads();
    float4 s = make_float4(0, 0, 0, 0), S1 = s, S2 = s;
    if (active) {
        s = sh[chunk];
        for (int64_t row = r0 + rsub; row < r1; row += a.RPB) {
            const float4 d = load_d<VEC>(a, row, chunk);
            const float tx = d.x - s.x, ty = d.y - s.y, tz = d.z - s.z, tw = d.w - s.w;
            S1.x += tx; S1.y += ty; S1.z += tz; S1.w += tw;
            S2.x += tx * tx; S2.y += ty * ty; S2.z += tz * tz; S2.w += tw * tw;
        }
    }
    __syncthreads();
    sh[t] = S1; sh2[t] = S2;
    __syncthreads();
    if (active && rsub == 0) {
        for (int r = 1; r < a.RPB; ++r) {
            const float4 u = sh[r * a.CH + chunk], w = sh2[r * a.CH + chunk];
            S1.x += u.x; S1.y += u.y; S1.z += u.z; S1.w += u.w;
            S2.x += w.x; S2.y += w.y; S2.z += w.z; S2.w += w.w;
        }
        const float n = (float)(r1 - r0);
        float4 m, q;
        m.x = s.x + S1.x / n; m.y = s.y + S1.y / n; m.z = s.z + S1.z / n; m.w = s.w + S1.w / n;
        q.x = S2.x - S1.x * S1.x / n; q.y = S2.y - S1.y * S1.y / n;
        q.z = S2.z - S1.z * S1.z / n; q.w = S2.w - S1.w * S1.w / n;
        const int Hp = a.CH * 4;
        float* pm = partial + ((int64_t)blockIdx.x * 2 + 0) * Hp + chunk * 4;
        float* pq = partial + ((int64_t)blockIdx.x * 2 + 1) * Hp + chunk * 4;
        *reinterpret_cast<float4*>(pm) = m;
        *reinterpret_cast<float4*>(pq) = q;
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
                                                           int64_t rows_per_block, int64_t N, int H, int Hp,
                                                           float eps, float momentum,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd) {
    __shared__ double sh[FIN_SL][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < H;
    double a1 = 0.0, a2 = 0.0, pivot = 0.0;
    if (ok) {
        pivot = (double)partial[c];
        const double n_full = (double)rows_per_block;
        const double n_tail = (double)(N - (int64_t)(nblocks - 1) * rows_per_block);
        // batches of 8 independent loads (clamped index, zero weight past the end) before the fp64 chain
        for (int b0 = sl; b0 < nblocks; b0 += 8 * FIN_SL) {
            float m[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min(b0 + u * FIN_SL, nblocks - 1);
                m[u] = partial[((int64_t)b * 2) * Hp + c];
                q[u] = partial[((int64_t)b * 2 + 1) * Hp + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * FIN_SL;
                const double nb = b >= nblocks ? 0.0 : (b == nblocks - 1 ? n_tail : n_full);
                const double d = (double)m[u] - pivot;
                a1 += nb * d;
                a2 += (b < nblocks ? (double)q[u] : 0.0) + nb * d * d;
            }
        }
    }
    const double s1 = fin_tree(sh, sl, cl, a1);
    const double s2 = fin_tree(sh, sl, cl, a2);
    if (sl != 0 || !ok) return;
    const double mean = pivot + s1 / (double)N;
    double m2 = s2 - s1 * s1 / (double)N;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / (double)N;
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = N > 1 ? m2 / (double)(N - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

struct NormArgs {
    const float* mean; const float* rstd_or_var; int rstd_is_var; float eps;
    const float* gamma; const float* beta;
    float* out; int64_t ldo;
};

__device__ __forceinline__ float4 ldcol4(const float* p, int c0, int H, float fill) {
    float4 v;
    v.x = (p && c0 + 0 < H) ? p[c0 + 0] : fill; v.y = (p && c0 + 1 < H) ? p[c0 + 1] : fill;
    v.z = (p && c0 + 2 < H) ? p[c0 + 2] : fill; v.w = (p && c0 + 3 < H) ? p[c0 + 3] : fill;
    return v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_ep_apply(const EpArgs a, const NormArgs n) {
    const int t = threadIdx.x;
    if (t >= a.RPB * a.CH) return;
    const int rsub = t / a.CH, chunk = t - rsub * a.CH, c0 = chunk * 4;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = min(a.N, r0 + a.rows_per_block);
    const float4 mu = ldcol4(n.mean, c0, a.H, 0.f);
    float4 rs = ldcol4(n.rstd_or_var, c0, a.H, 1.f);
    if (n.rstd_is_var) {
        rs.x = 1.f / sqrtf(rs.x + n.eps); rs.y = 1.f / sqrtf(rs.y + n.eps);
        rs.z = 1.f / sqrtf(rs.z + n.eps); rs.w = 1.f / sqrtf(rs.w + n.eps);
    }
    const float4 ga = ldcol4(n.gamma, c0, a.H, 1.f), be = ldcol4(n.beta, c0, a.H, 0.f);
    for (int64_t row = r0 + rsub; row < r1; row += a.RPB) {
        const float4 d = load_d<VEC>(a, row, chunk);
        float4 o;
        o.x = (d.x - mu.x) * rs.x * ga.x + be.x; o.y = (d.y - mu.y) * rs.y * ga.y + be.y;
        o.z = (d.z - mu.z) * rs.z * ga.z + be.z; o.w = (d.w - mu.w) * rs.w * ga.w + be.w;
        st4<VEC>(n.out, n.ldo, row, c0, a.H, o);
    }
}

// ---------------- backward ----------------
struct BwdArgs {
    const float* g; int64_t ldg;
    const float* mean; const float* rstd; const float* gamma;
    const float* sums;   // [2][Hp]: sum g, sum g*xhat (apply pass only)
    float* dy; int64_t lddy;
};

// partial layout: [block][2][Hp] = (sum g, sum g*xhat)
template <bool VEC>
__global__ __launch_bounds__(256) void k_ep_bwd_stats(const EpArgs a, const BwdArgs b, float* __restrict__ partial) {
    __shared__ float4 sh[256];
    __shared__ float4 sh2[256];
    const int t = threadIdx.x;
    const bool active = t < a.RPB * a.CH;
    const int rsub = t / a.CH, chunk = t - rsub * a.CH, c0 = chunk * 4;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = min(a.N, r0 + a.rows_per_block);
    float4 S1 = make_float4(0, 0, 0, 0), S2 = S1;
    if (active) {
        const float4 mu = ldcol4(b.mean, c0, a.H, 0.f), rs = ldcol4(b.rstd, c0, a.H, 0.f);
        for (int64_t row = r0 + rsub; row < r1; row += a.RPB) {
            const float4 d = load_d<VEC>(a, row, chunk);
            const float4 g = ld4<VEC>(b.g, b.ldg, row, c0, a.H);
            S1.x += g.x; S1.y += g.y; S1.z += g.z; S1.w += g.w;
            S2.x += g.x * ((d.x - mu.x) * rs.x); S2.y += g.y * ((d.y - mu.y) * rs.y);
            S2.z += g.z * ((d.z - mu.z) * rs.z); S2.w += g.w * ((d.w - mu.w) * rs.w);
        }
    }
    sh[t] = S1; sh2[t] = S2;
    __syncthreads();
    if (active && rsub == 0) {
        for (int r = 1; r < a.RPB; ++r) {
            const float4 u = sh[r * a.CH + chunk], w = sh2[r * a.CH + chunk];
            S1.x += u.x; S1.y += u.y; S1.z += u.z; S1.w += u.w;
            S2.x += w.x; S2.y += w.y; S2.z += w.z; S2.w += w.w;
        }
        const int Hp = a.CH * 4;
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * Hp + c0) = S1;
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * Hp + c0) = S2;
    }
}

__global__ __launch_bounds__(1024) void k_ep_bwd_finalize(const float* __restrict__ partial, int nblocks, int H,
                                                         int Hp, float* __restrict__ sums,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[FIN_SL][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < H) {
        for (int b0 = sl; b0 < nblocks; b0 += 8 * FIN_SL) {
            float m[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min(b0 + u * FIN_SL, nblocks - 1);
                m[u] = partial[((int64_t)b * 2 + 0) * Hp + c];
                q[u] = partial[((int64_t)b * 2 + 1) * Hp + c];
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
    if (sl != 0 || c >= Hp) return;
    sums[c] = (float)s1;
    sums[Hp + c] = (float)s2;
    if (c < H) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_ep_bwd_apply(const EpArgs a, const BwdArgs b) {
    const int t = threadIdx.x;
    if (t >= a.RPB * a.CH) return;
    const int rsub = t / a.CH, chunk = t - rsub * a.CH, c0 = chunk * 4;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = min(a.N, r0 + a.rows_per_block);
    const int Hp = a.CH * 4;
    const float invn = 1.f / (float)a.N;
    const float4 mu = ldcol4(b.mean, c0, a.H, 0.f), rs = ldcol4(b.rstd, c0, a.H, 0.f);
    const float4 ga = ldcol4(b.gamma, c0, a.H, 1.f);
    const float4 sg = *reinterpret_cast<const float4*>(b.sums + c0);
    const float4 sgx = *reinterpret_cast<const float4*>(b.sums + Hp + c0);
    for (int64_t row = r0 + rsub; row < r1; row += a.RPB) {
        float4 y, k;
        const float4 d = load_d<VEC>(a, row, chunk, &y, &k);
        const float4 g = ld4<VEC>(b.g, b.ldg, row, c0, a.H);
        float4 o;
#define HGV_EP_BWD(f)                                                                              \
        {                                                                                          \
            const float xh = (d.f - mu.f) * rs.f;                                                  \
            float dd = ga.f * rs.f * (g.f - sg.f * invn - xh * (sgx.f * invn));                   \
            dd *= k.f;                                                                             \
            if (a.relu && !(y.f > 0.f)) dd = 0.f;                                                  \
            o.f = dd;                                                                              \
        }
        HGV_EP_BWD(x) HGV_EP_BWD(y) HGV_EP_BWD(z) HGV_EP_BWD(w)
#undef HGV_EP_BWD
        st4<VEC>(b.dy, b.lddy, row, c0, a.H, o);
    }
}

bool vec4_ok(const float* p, int64_t ld, int Hp) { return p == nullptr || (hgv_aligned(p, 16) && ld % 4 == 0 && ld >= Hp); }

int make_args(EpArgs& a, const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
              uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int* nblocks) {
    HGV_REQUIRE(N > 0 && H > 0 && H <= 1024, HGV_EINVAL, "bn_act: need N > 0 and 0 < H <= 1024 (N=%lld H=%d)", (long long)N, H);
    HGV_REQUIRE(y && ldy >= H, HGV_EINVAL, "bn_act: bad input buffer");
    HGV_REQUIRE(p >= 0.f && p < 1.f, HGV_EINVAL, "bn_act: dropout p=%f outside [0,1)", p);
    a.y = y; a.ldy = ldy; a.N = N; a.H = H; a.CH = (H + 3) / 4; a.RPB = 256 / a.CH;
    a.relu = relu; a.p = p; a.scale = 1.0f / (float)(1.0 - (double)p);
    a.mask = mask; a.seed = seed; a.offset = offset; a.offset_dev = offset_dev;
    a.mode = (p > 0.f) ? (mask ? 1 : 2) : 0;
    // row ranges: enough workgroups to fill the chip (8 per CU), at least 8 rows per thread
    int64_t nb = hgv_cdiv(N, (int64_t)a.RPB * 8);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    a.rows_per_block = hgv_cdiv(N, nb);
    *nblocks = (int)hgv_cdiv(N, a.rows_per_block);
    return HGV_OK;
}

}  // namespace

extern "C" size_t hgv_bn_act_ws(int64_t N, int H) {
    (void)N;
    const size_t Hp = (size_t)((H + 3) / 4) * 4;
    return hgv_align256(2048 * 2 * Hp * sizeof(float)) + hgv_align256(2 * Hp * sizeof(float));
}

// Training-mode forward.  mask: explicit keep mask [N,H] (uint8, contiguous) or null; with
// p > 0 and no mask the keep decisions come from Philox4x32-10 keyed by (seed, offset, row,
// chunk) and the backward pass regenerates them from the same (seed, offset).
extern "C" int hgv_bn_act_fwd(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
                              uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma,
                              const float* beta, float eps,
                              float* running_mean, float* running_var, float momentum, float* save_mean,
                              float* save_rstd, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(out && ldo >= H && save_mean && save_rstd && ws, HGV_EINVAL, "bn_act_fwd: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_fwd: workspace too small or misaligned");
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    const bool vec = vec4_ok(y, ldy, Hp) && vec4_ok(out, ldo, Hp);
    if (vec) hipLaunchKernelGGL(k_ep_stats<true>, dim3(nb), dim3(256), 0, stream, a, partial);
    else hipLaunchKernelGGL(k_ep_stats<false>, dim3(nb), dim3(256), 0, stream, a, partial);
    hipLaunchKernelGGL(k_ep_stats_finalize, dim3((H + 15) / 16), dim3(1024), 0, stream, partial, nb, a.rows_per_block, N,
                       H, Hp, eps, momentum, running_mean, running_var, save_mean, save_rstd);
    NormArgs n{save_mean, save_rstd, 0, eps, gamma, beta, out, ldo};
    if (vec) hipLaunchKernelGGL(k_ep_apply<true>, dim3(nb), dim3(256), 0, stream, a, n);
    else hipLaunchKernelGGL(k_ep_apply<false>, dim3(nb), dim3(256), 0, stream, a, n);
    HGV_CHECK_LAUNCH("bn_act_fwd");
    return HGV_OK;
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
    NormArgs n{running_mean, running_var, 1, eps, gamma, beta, out, ldo};
    if (vec4_ok(y, ldy, Hp) && vec4_ok(out, ldo, Hp)) hipLaunchKernelGGL(k_ep_apply<true>, dim3(nb), dim3(256), 0, stream, a, n);
    else hipLaunchKernelGGL(k_ep_apply<false>, dim3(nb), dim3(256), 0, stream, a, n);
    HGV_CHECK_LAUNCH("bn_act_eval");
    return HGV_OK;
}

// Backward of the training-mode forward: dy (w.r.t. the pre-activation), dgamma, dbeta.
extern "C" int hgv_bn_act_bwd(const float* g, int64_t ldg, const float* y, int64_t ldy, int64_t N, int H, int relu,
                              float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                              const uint64_t* offset_dev, const float* gamma,
                              const float* save_mean, const float* save_rstd, float* dy, int64_t lddy,
                              float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t stream) {
    EpArgs a; int nb;
    int rc = make_args(a, y, ldy, N, H, relu, p, mask, seed, offset, offset_dev, &nb);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(g && ldg >= H && dy && lddy >= H && save_mean && save_rstd && ws, HGV_EINVAL, "bn_act_bwd: null/short buffer");
    HGV_REQUIRE(ws_bytes >= hgv_bn_act_ws(N, H) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "bn_act_bwd: workspace too small or misaligned");
    const int Hp = a.CH * 4;
    float* partial = (float*)ws;
    float* sums = (float*)((char*)ws + hgv_align256(2048 * 2 * (size_t)Hp * sizeof(float)));
    BwdArgs b{g, ldg, save_mean, save_rstd, gamma, sums, dy, lddy};
    const bool vec = vec4_ok(y, ldy, Hp) && vec4_ok(g, ldg, Hp) && vec4_ok(dy, lddy, Hp);
    if (vec) hipLaunchKernelGGL(k_ep_bwd_stats<true>, dim3(nb), dim3(256), 0, stream, a, b, partial);
    else hipLaunchKernelGGL(k_ep_bwd_stats<false>, dim3(nb), dim3(256), 0, stream, a, b, partial);
    hipLaunchKernelGGL(k_ep_bwd_finalize, dim3((Hp + 15) / 16), dim3(1024), 0, stream, partial, nb, H, Hp, sums, dgamma, dbeta);
    if (vec) hipLaunchKernelGGL(k_ep_bwd_apply<true>, dim3(nb), dim3(256), 0, stream, a, b);
    else hipLaunchKernelGGL(k_ep_bwd_apply<false>, dim3(nb), dim3(256), 0, stream, a, b);
    HGV_CHECK_LAUNCH("bn_act_bwd");
    return HGV_OK;
}
