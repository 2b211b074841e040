// Single-head graph attention over the by-target CSR of the loop-augmented graph (edges without self
// loops, then one loop per node): edge softmax, attention dropout and the weighted neighbour sum in ONE
// kernel per direction; nothing per-edge is written to memory.
//
// Reference: GATConv(hidden, hidden, heads=1, dropout=0.49) at
// gvae/gvae_skipcon_attention.py:68,109 (PyG defaults negative_slope=0.2, add_self_loops):
//   e_k      = leaky_relu(a_src[col_k] + a_dst[i], slope)            k in row i
//   alpha_k  = exp(e_k - max_row e) / (sum_row exp(e - max) + 1e-16)
//   alpha'_k = dropout(alpha_k)        (mask indexed by the position of the edge in the edge LIST)
//   out_i    = sum_k alpha'_k * xw[col_k] + bias
//
// Forward (k_gat_fwd): a group of L lanes owns a target row (the mapping of the aggregation kernel: lane l
// keeps 16-byte feature chunks l, l + L, ...).  The logits of a row only touch col / a_src (4 bytes per
// edge), so the row maximum and the normaliser take two cheap sweeps over them; the third sweep forms the
// weights L edges at a time (one edge per lane), broadcasts them with ds_bpermute and accumulates the
// gathered rows of xw.  Per row it keeps (max, normaliser) for the backward pass.
//
// Backward: with dad_k = <g_i, xw[col_k]>, S_i = sum_k alpha'_k dad_k and lk_k = d leaky / d raw,
//   d logit_k = (alpha'_k dad_k - alpha_k S_i) lk_k,   d a_dst[i] = sum_k d logit_k,   d a_src[j] = sum over
//   the out-edges of j,   d xw[j] = sum over the out-edges (j -> i) of alpha'_k g_i.
// k_gat_bwd_rows walks the by-target CSR (gathers xw rows against the row's g_i held in registers) and
// writes S_i and d a_dst[i], both as sums of differences from a value near S_i (see there).  k_gat_bwd_cols walks the by-source
// CSR: it holds xw[j] in registers, gathers g_i once per edge for both the dot product and the weighted
// sum, and recomputes alpha / the keep decision of every edge from (a_src[j], a_dst[i], max_i, normaliser_i,
// edge id), so no per-edge array and no permutation between the two CSR orders is needed.  No atomics.
#include "common.h"

namespace {

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

struct GatArgs {
    const int32_t* rowptr; const int32_t* col; const int32_t* perm; int64_t N;
    const float* a_src; int64_t lds_; const float* a_dst; int64_t ldd_;
    float slope, p, scale; const uint8_t* mask; int mode;       // mode 0: no dropout, 1: explicit mask, 2: Philox
    uint64_t seed, offset; const uint64_t* offset_dev;
    int C, nchunks, lshift;
    const float* xw; int64_t ldx;
    const float* g; int64_t ldg;
    const float* bias;
    float* out; int64_t ldo;            // forward: out; backward over sources: d xw
    float* row_max; float* row_den;     // per target row: max logit, sum exp + 1e-16
    float* S;                           // per target row i: S_i = sum_k alpha'_k dad_k as r_i (S[i]) + V_i (S[N + i])
    float* da; int64_t ld_da;           // rows kernel: d a_dst; cols kernel: d a_src (element stride ld_da)
    const float* da_other;              // cols kernel: d a_dst of the rows kernel (same stride), for the logit projection term
    const float* att_src; const float* att_dst;   // cols kernel, nullable: a_src = <xw, att_src>, a_dst = <xw, att_dst>; then
                                        // d xw[j] += d a_src[j] att_src + d a_dst[j] att_dst (the gradient through the logits)
    int long_threshold;                 // > 0: rows with more entries than this are left to the k_gat_*_long kernels
};

// keep multiplier (0 or 1/(1-p)) of the edge with edge-list position eid
__device__ __forceinline__ float edge_keep(const GatArgs& a, uint32_t eid, uint64_t off) {
    if (a.mode == 0) return 1.f;
    if (a.mode == 1) return a.mask[eid] ? a.scale : 0.f;
    const HgvPhilox ph(a.seed);
    const uint4 r = ph(eid, 0x6A7u, (uint32_t)off, (uint32_t)(off >> 32));
    return hgv_u01(r.x) >= a.p ? a.scale : 0.f;
}

__device__ __forceinline__ float grp_sum(float v, int L) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        if (o < L) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float grp_max(float v, int L) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        if (o < L) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void axpy4(float4& acc, float w, const float4& v) {
    acc.x = __fadd_rn(acc.x, __fmul_rn(w, v.x)); acc.y = __fadd_rn(acc.y, __fmul_rn(w, v.y));
    acc.z = __fadd_rn(acc.z, __fmul_rn(w, v.z)); acc.w = __fadd_rn(acc.w, __fmul_rn(w, v.w));
}

// row of the thread's group (two waves per workgroup, XCD-aware block order); false when past the end
__device__ __forceinline__ bool group_row(const GatArgs& a, int64_t& row, int& lr, int& L) {
    const int lane = threadIdx.x & 63;
    L = 1 << a.lshift;
    lr = lane & (L - 1);
    const int64_t wave = (int64_t)hgv_xcd_remap(blockIdx.x, gridDim.x) * 2 + (threadIdx.x >> 6);
    row = wave * (64 >> a.lshift) + (lane >> a.lshift);
    return row < a.N;
}

// the lane's CPL chunks of row `r` of `base`; columns past C and chunks past the row read as zeros
template <int CPL>
__device__ __forceinline__ void load_own_row(const float* base, int64_t ld, int64_t r, int C, int nchunks, int lr, int L,
                                             float4 (&v)[CPL], int (&coff)[CPL], bool (&owns)[CPL]) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lr + q * L;
        owns[q] = c < nchunks;
        coff[q] = (owns[q] ? c : 0) * 4;
        float4 t = *reinterpret_cast<const float4*>(base + r * ld + coff[q]);
        if (!owns[q]) t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (coff[q] + 1 >= C) t.y = 0.f;
        if (coff[q] + 2 >= C) t.z = 0.f;
        if (coff[q] + 3 >= C) t.w = 0.f;
        v[q] = t;
    }
}

template <int CPL>
__global__ __launch_bounds__(128) void k_gat_fwd(const GatArgs a) {
    int64_t row; int lr, L;
    if (!group_row(a, row, lr, L)) return;
    const int start = a.rowptr[row], end = a.rowptr[row + 1];
    if (a.long_threshold > 0 && end - start > a.long_threshold) return;      // the whole group leaves: see k_gat_fwd_long
    const float ad = a.a_dst[row * a.ldd_];
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    float m = -INFINITY;
    for (int k = start + lr; k < end; k += L) m = fmaxf(m, leaky(a.a_src[(int64_t)a.col[k] * a.lds_] + ad, a.slope));
    m = grp_max(m, L);
    float s = 0.f;
    for (int k = start + lr; k < end; k += L) s += expf(leaky(a.a_src[(int64_t)a.col[k] * a.lds_] + ad, a.slope) - m);
    s = grp_sum(s, L) + 1e-16f;

    bool owns[CPL]; int coff[CPL]; float4 acc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lr + q * L;
        owns[q] = c < a.nchunks;
        coff[q] = (owns[q] ? c : a.nchunks - 1) * 4;
        acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int base = start; base < end; base += L) {
        const int k = base + lr;
        int my_col = 0; float my_w = 0.f;
        if (k < end) {
            my_col = a.col[k];
            const float al = expf(leaky(a.a_src[(int64_t)my_col * a.lds_] + ad, a.slope) - m) / s;
            my_w = al * edge_keep(a, (uint32_t)a.perm[k], off);
        }
        const int cnt = min(L, end - base);
        for (int j = 0; j < cnt; j += 2) {                 // two neighbour rows in flight; adds stay in segment order
            const int j1 = min(j + 1, cnt - 1);
            const float* r0 = a.xw + (int64_t)__shfl(my_col, j, L) * a.ldx;
            const float* r1 = a.xw + (int64_t)__shfl(my_col, j1, L) * a.ldx;
            const float w0 = __shfl(my_w, j, L), w1 = j + 1 < cnt ? __shfl(my_w, j1, L) : 0.f;
            float4 v0[CPL], v1[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                v0[q] = *reinterpret_cast<const float4*>(r0 + coff[q]);
                v1[q] = *reinterpret_cast<const float4*>(r1 + coff[q]);
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) axpy4(acc[q], w0, v0[q]);
            if (j + 1 < cnt) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) axpy4(acc[q], w1, v1[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        if (owns[q]) {
            float4 o = acc[q];
            if (a.bias) {
                const int c0 = coff[q];
                o.x += a.bias[c0];
                if (c0 + 1 < a.C) o.y += a.bias[c0 + 1];
                if (c0 + 2 < a.C) o.z += a.bias[c0 + 2];
                if (c0 + 3 < a.C) o.w += a.bias[c0 + 3];
            }
            if (coff[q] + 1 >= a.C) o.y = 0.f;
            if (coff[q] + 2 >= a.C) o.z = 0.f;
            if (coff[q] + 3 >= a.C) o.w = 0.f;
            *reinterpret_cast<float4*>(a.out + row * a.ldo + coff[q]) = o;
        }
    }
    if (lr == 0) { a.row_max[row] = m; a.row_den[row] = s; }
}

// by-target rows: S_i = sum_k alpha'_k <g_i, xw[col_k]>,  d a_dst[i] = sum_k (alpha'_k dad_k - alpha_k S_i) lk_k
template <int CPL>
__global__ __launch_bounds__(128) void k_gat_bwd_rows(const GatArgs a) {
    int64_t row; int lr, L;
    if (!group_row(a, row, lr, L)) return;
    const int start = a.rowptr[row], end = a.rowptr[row + 1];
    if (a.long_threshold > 0 && end - start > a.long_threshold) return;
    float4 gr[CPL]; int coff[CPL]; bool owns[CPL];
    load_own_row<CPL>(a.g, a.ldg, row, a.C, a.nchunks, lr, L, gr, coff, owns);
    const float ad = a.a_dst[row * a.ldd_], m = a.row_max[row], s = a.row_den[row];
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    // With t_k = m_k dad_k - r (m_k the keep multiplier, r a value near the row's S taken from the first L edges):
    //   S = r + sum_k alpha_k t_k,   d a_dst[i] = sum_k alpha_k lk_k (m_k dad_k - S) = U - V T2   (sum_k alpha_k = 1),
    //   U = sum alpha lk t, V = sum alpha t, T2 = sum alpha lk.
    // All sums run over differences from r: when the softmax is sharp (or all dad_k alike) they are small, while the plain
    // sum alpha' dad lk - S sum alpha lk subtracts two numbers of the size of dad itself.
    float U = 0.f, V = 0.f, T2 = 0.f, r = 0.f;
    for (int base = start; base < end; base += L) {
        const int k = base + lr;
        int my_col = 0; float al = 0.f, mk = 0.f, lk = 0.f;
        if (k < end) {
            my_col = a.col[k];
            const float raw = a.a_src[(int64_t)my_col * a.lds_] + ad;
            al = expf(leaky(raw, a.slope) - m) / s;
            mk = edge_keep(a, (uint32_t)a.perm[k], off);
            lk = raw > 0.f ? 1.f : a.slope;
        }
        const int cnt = min(L, end - base);
        float my_dad = 0.f;
        for (int j = 0; j < cnt; j += 2) {
            const int j1 = min(j + 1, cnt - 1);
            const float* r0 = a.xw + (int64_t)__shfl(my_col, j, L) * a.ldx;
            const float* r1 = a.xw + (int64_t)__shfl(my_col, j1, L) * a.ldx;
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                d0 += dot4(gr[q], *reinterpret_cast<const float4*>(r0 + coff[q]));
                d1 += dot4(gr[q], *reinterpret_cast<const float4*>(r1 + coff[q]));
            }
            d0 = grp_sum(d0, L); d1 = grp_sum(d1, L);
            if (lr == j) my_dad = d0;
            if (lr == j + 1) my_dad = d1;
        }
        if (base == start) {
            const float num = grp_sum(al * mk * my_dad, L), den = grp_sum(al * mk, L);
            r = den > 0.f ? num / den : 0.f;
        }
        const float t = k < end ? mk * my_dad - r : 0.f;
        U += al * lk * t;
        V += al * t;
        T2 += al * lk;
    }
    U = grp_sum(U, L); V = grp_sum(V, L); T2 = grp_sum(T2, L);
    if (lr == 0) { a.S[row] = r; a.S[a.N + row] = V; a.da[row * a.ld_da] = U - V * T2; }
}

// by-source rows (rowptr / col / perm of the transposed CSR): d xw[j] and d a_src[j]
template <int CPL>
__global__ __launch_bounds__(128) void k_gat_bwd_cols(const GatArgs a) {
    int64_t row; int lr, L;
    if (!group_row(a, row, lr, L)) return;
    const int start = a.rowptr[row], end = a.rowptr[row + 1];
    if (a.long_threshold > 0 && end - start > a.long_threshold) return;
    float4 xr[CPL], acc[CPL]; int coff[CPL]; bool owns[CPL];
    load_own_row<CPL>(a.xw, a.ldx, row, a.C, a.nchunks, lr, L, xr, coff, owns);
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float as = a.a_src[row * a.lds_];
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    float dsum = 0.f;
    for (int base = start; base < end; base += L) {
        const int t = base + lr;
        int my_col = 0; float al = 0.f, ap = 0.f, mk = 0.f, lk = 0.f, ri = 0.f, Vi = 0.f;
        if (t < end) {
            my_col = a.col[t];                                    // target i of this out-edge
            const float raw = as + a.a_dst[(int64_t)my_col * a.ldd_];
            al = expf(leaky(raw, a.slope) - a.row_max[my_col]) / a.row_den[my_col];
            mk = edge_keep(a, (uint32_t)a.perm[t], off);
            ap = al * mk;
            lk = raw > 0.f ? 1.f : a.slope;
            ri = a.S[my_col]; Vi = a.S[a.N + my_col];
        }
        const int cnt = min(L, end - base);
        float my_dad = 0.f;
        for (int j = 0; j < cnt; j += 2) {
            const int j1 = min(j + 1, cnt - 1);
            const float* r0 = a.g + (int64_t)__shfl(my_col, j, L) * a.ldg;
            const float* r1 = a.g + (int64_t)__shfl(my_col, j1, L) * a.ldg;
            const float w0 = __shfl(ap, j, L), w1 = j + 1 < cnt ? __shfl(ap, j1, L) : 0.f;
            float4 v0[CPL], v1[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                v0[q] = *reinterpret_cast<const float4*>(r0 + coff[q]);
                v1[q] = *reinterpret_cast<const float4*>(r1 + coff[q]);
            }
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { d0 += dot4(xr[q], v0[q]); d1 += dot4(xr[q], v1[q]); }
#pragma unroll
            for (int q = 0; q < CPL; ++q) axpy4(acc[q], w0, v0[q]);
            if (j + 1 < cnt) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) axpy4(acc[q], w1, v1[q]);
            }
            d0 = grp_sum(d0, L); d1 = grp_sum(d1, L);
            if (lr == j) my_dad = d0;
            if (lr == j + 1) my_dad = d1;
        }
        dsum += al * ((mk * my_dad - ri) - Vi) * lk;      // S_i = r_i + V_i kept in two parts: both differences are small where the softmax is sharp
    }
    dsum = grp_sum(dsum, L);
    const float dd = a.att_src ? a.da_other[row * a.ld_da] : 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        if (owns[q]) {
            float4 o = acc[q];
            if (a.att_src) {
                const float4 s4 = *reinterpret_cast<const float4*>(a.att_src + coff[q]);
                const float4 d4 = *reinterpret_cast<const float4*>(a.att_dst + coff[q]);
                o.x += dsum * s4.x + dd * d4.x; o.y += dsum * s4.y + dd * d4.y;
                o.z += dsum * s4.z + dd * d4.z; o.w += dsum * s4.w + dd * d4.w;
            }
            if (coff[q] + 1 >= a.C) o.y = 0.f;
            if (coff[q] + 2 >= a.C) o.z = 0.f;
            if (coff[q] + 3 >= a.C) o.w = 0.f;
            *reinterpret_cast<float4*>(a.out + row * a.ldo + coff[q]) = o;
        }
    }
    if (lr == 0) a.da[row * a.ld_da] = dsum;
}

// ---- rows far longer than the average ("hubs") ------------------------------------------------------------------------
// A lane group walks its row alone, so one row of a few thousand entries outlasts the rest of the launch (the aggregation
// kernel has the same problem and the same cure).  Such rows are listed once per CSR (hgv_csr_long_rows) and handled by a
// whole 512-thread workgroup each: wave w of 8 takes the w-th eighth of the row's segment with the whole wave as the
// lane group (lane l owns the chunks l, l + 64, ...), and the eight partial results are combined in wave order through
// LDS - no atomics, the same result on every run; for these rows the summation is blocked, not sequential.
constexpr int GAT_LONG_WAVES = 8;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// the same value in every thread of the workgroup: the partials of the waves combined in wave order
__device__ __forceinline__ float wg_combine(float v, float* red, int w, int lane, bool is_max) {
    __syncthreads();                                   // the previous use of red is over
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int q = 1; q < GAT_LONG_WAVES; ++q) t = is_max ? fmaxf(t, red[q]) : t + red[q];
    return t;
}

template <int CPL>
__global__ __launch_bounds__(64 * GAT_LONG_WAVES) void k_gat_fwd_long(const GatArgs a, const int32_t* __restrict__ rows,
                                                                      const int32_t* __restrict__ count, int cap) {
    __shared__ float4 sh[GAT_LONG_WAVES * 64 * CPL];
    __shared__ float red[GAT_LONG_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_long = min(count[0], cap);
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    for (int idx = blockIdx.x; idx < n_long; idx += gridDim.x) {
        const int64_t row = rows[idx];
        const int start = a.rowptr[row], end = a.rowptr[row + 1], len = end - start;
        const int s0 = start + (int)((int64_t)len * w / GAT_LONG_WAVES), s1 = start + (int)((int64_t)len * (w + 1) / GAT_LONG_WAVES);
        const float ad = a.a_dst[row * a.ldd_];
        float m = -INFINITY;
        for (int k = s0 + lane; k < s1; k += 64) m = fmaxf(m, leaky(a.a_src[(int64_t)a.col[k] * a.lds_] + ad, a.slope));
        m = wg_combine(wave_max(m), red, w, lane, true);
        float s = 0.f;
        for (int k = s0 + lane; k < s1; k += 64) s += expf(leaky(a.a_src[(int64_t)a.col[k] * a.lds_] + ad, a.slope) - m);
        s = wg_combine(wave_sum(s), red, w, lane, false) + 1e-16f;
        bool owns[CPL]; int coff[CPL]; float4 acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane + q * 64;
            owns[q] = c < a.nchunks;
            coff[q] = (owns[q] ? c : a.nchunks - 1) * 4;
            acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int base = s0; base < s1; base += 64) {
            const int k = base + lane;
            int my_col = 0; float my_w = 0.f;
            if (k < s1) {
                my_col = a.col[k];
                const float al = expf(leaky(a.a_src[(int64_t)my_col * a.lds_] + ad, a.slope) - m) / s;
                my_w = al * edge_keep(a, (uint32_t)a.perm[k], off);
            }
            const int cnt = min(64, s1 - base);
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {                 // four neighbour rows in flight: the row is all this workgroup has
                float4 v[4][CPL]; float wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* r = a.xw + (int64_t)__shfl(my_col, j + u, 64) * a.ldx;
                    wv[u] = __shfl(my_w, j + u, 64);
#pragma unroll
                    for (int q = 0; q < CPL; ++q) v[u][q] = *reinterpret_cast<const float4*>(r + coff[q]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < CPL; ++q) axpy4(acc[q], wv[u], v[u][q]);
            }
            for (; j < cnt; ++j) {
                const float* r = a.xw + (int64_t)__shfl(my_col, j, 64) * a.ldx;
                const float wj = __shfl(my_w, j, 64);
#pragma unroll
                for (int q = 0; q < CPL; ++q) axpy4(acc[q], wj, *reinterpret_cast<const float4*>(r + coff[q]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPL; ++q) sh[(w * CPL + q) * 64 + lane] = acc[q];
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                float4 o = sh[q * 64 + lane];
                for (int ww = 1; ww < GAT_LONG_WAVES; ++ww) {
                    const float4 t = sh[(ww * CPL + q) * 64 + lane];
                    o.x = __fadd_rn(o.x, t.x); o.y = __fadd_rn(o.y, t.y); o.z = __fadd_rn(o.z, t.z); o.w = __fadd_rn(o.w, t.w);
                }
                if (owns[q]) {
                    const int c0 = coff[q];
                    if (a.bias) {
                        o.x += a.bias[c0];
                        if (c0 + 1 < a.C) o.y += a.bias[c0 + 1];
                        if (c0 + 2 < a.C) o.z += a.bias[c0 + 2];
                        if (c0 + 3 < a.C) o.w += a.bias[c0 + 3];
                    }
                    if (c0 + 1 >= a.C) o.y = 0.f;
                    if (c0 + 2 >= a.C) o.z = 0.f;
                    if (c0 + 3 >= a.C) o.w = 0.f;
                    *reinterpret_cast<float4*>(a.out + row * a.ldo + c0) = o;
                }
            }
            if (lane == 0) { a.row_max[row] = m; a.row_den[row] = s; }
        }
    }
}

// <g_i, xw row> for the row whose chunks the lanes of a wave hold in gr; the same value in every lane
template <int CPL>
__device__ __forceinline__ float wave_dot(const float4 (&gr)[CPL], const float* r, const int (&coff)[CPL]) {
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) d += dot4(gr[q], *reinterpret_cast<const float4*>(r + coff[q]));
    return wave_sum(d);
}

// by-target hub rows.  Every wave takes the pivot of its own part (the alpha'-weighted mean of the dot products of the
// part's first 64 edges) and sums differences from it: with A_w = sum alpha, T2_w = sum alpha lk, V_w = sum alpha t, U_w =
// sum alpha lk t (t = m dad - r_w) over the part, the row's sums relative to r = r_0 are V = sum_w V_w + A_w (r_w - r) and
// U = sum_w U_w + T2_w (r_w - r).
template <int CPL>
__global__ __launch_bounds__(64 * GAT_LONG_WAVES) void k_gat_bwd_rows_long(const GatArgs a, const int32_t* __restrict__ rows,
                                                                           const int32_t* __restrict__ count, int cap) {
    __shared__ float part[GAT_LONG_WAVES][5];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_long = min(count[0], cap);
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    for (int idx = blockIdx.x; idx < n_long; idx += gridDim.x) {
        const int64_t row = rows[idx];
        float4 gr[CPL]; int coff[CPL]; bool owns[CPL];
        load_own_row<CPL>(a.g, a.ldg, row, a.C, a.nchunks, lane, 64, gr, coff, owns);
        const int start = a.rowptr[row], end = a.rowptr[row + 1], len = end - start;
        const int s0 = start + (int)((int64_t)len * w / GAT_LONG_WAVES), s1 = start + (int)((int64_t)len * (w + 1) / GAT_LONG_WAVES);
        const float ad = a.a_dst[row * a.ldd_], m = a.row_max[row], s = a.row_den[row];
        float U = 0.f, V = 0.f, T2 = 0.f, A = 0.f, r = 0.f;
        for (int base = s0; base < s1; base += 64) {
            const int k = base + lane;
            int my_col = 0; float al = 0.f, mk = 0.f, lk = 0.f;
            if (k < s1) {
                my_col = a.col[k];
                const float raw = a.a_src[(int64_t)my_col * a.lds_] + ad;
                al = expf(leaky(raw, a.slope) - m) / s;
                mk = edge_keep(a, (uint32_t)a.perm[k], off);
                lk = raw > 0.f ? 1.f : a.slope;
            }
            const int cnt = min(64, s1 - base);
            float my_dad = 0.f;
            for (int j = 0; j < cnt; ++j) {
                const float d = wave_dot<CPL>(gr, a.xw + (int64_t)__shfl(my_col, j, 64) * a.ldx, coff);
                if (lane == j) my_dad = d;
            }
            if (base == s0) {
                const float num = wave_sum(al * mk * my_dad), den = wave_sum(al * mk);
                r = den > 0.f ? num / den : 0.f;
            }
            const float t = k < s1 ? mk * my_dad - r : 0.f;
            U += al * lk * t;
            V += al * t;
            T2 += al * lk;
            A += al;
        }
        U = wave_sum(U); V = wave_sum(V); T2 = wave_sum(T2); A = wave_sum(A);
        __syncthreads();
        if (lane == 0) { part[w][0] = r; part[w][1] = U; part[w][2] = V; part[w][3] = T2; part[w][4] = A; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float r0 = part[0][0];
            float Ut = 0.f, Vt = 0.f, Tt = 0.f;
            for (int q = 0; q < GAT_LONG_WAVES; ++q) {
                const float dr = part[q][0] - r0;
                Ut += part[q][1] + part[q][3] * dr;
                Vt += part[q][2] + part[q][4] * dr;
                Tt += part[q][3];
            }
            a.S[row] = r0; a.S[a.N + row] = Vt; a.da[row * a.ld_da] = Ut - Vt * Tt;
        }
    }
}

// by-source hub rows (rowptr / col / perm of the transposed CSR)
template <int CPL>
__global__ __launch_bounds__(64 * GAT_LONG_WAVES) void k_gat_bwd_cols_long(const GatArgs a, const int32_t* __restrict__ rows,
                                                                           const int32_t* __restrict__ count, int cap) {
    __shared__ float4 sh[GAT_LONG_WAVES * 64 * CPL];
    __shared__ float red[GAT_LONG_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_long = min(count[0], cap);
    const uint64_t off = a.offset + (a.offset_dev ? a.offset_dev[0] : 0ull);
    for (int idx = blockIdx.x; idx < n_long; idx += gridDim.x) {
        const int64_t row = rows[idx];
        float4 xr[CPL], acc[CPL]; int coff[CPL]; bool owns[CPL];
        load_own_row<CPL>(a.xw, a.ldx, row, a.C, a.nchunks, lane, 64, xr, coff, owns);
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int start = a.rowptr[row], end = a.rowptr[row + 1], len = end - start;
        const int s0 = start + (int)((int64_t)len * w / GAT_LONG_WAVES), s1 = start + (int)((int64_t)len * (w + 1) / GAT_LONG_WAVES);
        const float as = a.a_src[row * a.lds_];
        float dsum = 0.f;
        for (int base = s0; base < s1; base += 64) {
            const int t = base + lane;
            int my_col = 0; float al = 0.f, ap = 0.f, mk = 0.f, lk = 0.f, ri = 0.f, Vi = 0.f;
            if (t < s1) {
                my_col = a.col[t];
                const float raw = as + a.a_dst[(int64_t)my_col * a.ldd_];
                al = expf(leaky(raw, a.slope) - a.row_max[my_col]) / a.row_den[my_col];
                mk = edge_keep(a, (uint32_t)a.perm[t], off);
                ap = al * mk;
                lk = raw > 0.f ? 1.f : a.slope;
                ri = a.S[my_col]; Vi = a.S[a.N + my_col];
            }
            const int cnt = min(64, s1 - base);
            float my_dad = 0.f;
            for (int j = 0; j < cnt; ++j) {
                const float* r0 = a.g + (int64_t)__shfl(my_col, j, 64) * a.ldg;
                const float w0 = __shfl(ap, j, 64);
                float4 v0[CPL];
                float d0 = 0.f;
#pragma unroll
                for (int q = 0; q < CPL; ++q) { v0[q] = *reinterpret_cast<const float4*>(r0 + coff[q]); d0 += dot4(xr[q], v0[q]); }
#pragma unroll
                for (int q = 0; q < CPL; ++q) axpy4(acc[q], w0, v0[q]);
                d0 = wave_sum(d0);
                if (lane == j) my_dad = d0;
            }
            dsum += al * ((mk * my_dad - ri) - Vi) * lk;
        }
        dsum = wg_combine(wave_sum(dsum), red, w, lane, false);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPL; ++q) sh[(w * CPL + q) * 64 + lane] = acc[q];
        __syncthreads();
        if (w == 0) {
            const float dd = a.att_src ? a.da_other[row * a.ld_da] : 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                float4 o = sh[q * 64 + lane];
                for (int ww = 1; ww < GAT_LONG_WAVES; ++ww) {
                    const float4 t = sh[(ww * CPL + q) * 64 + lane];
                    o.x = __fadd_rn(o.x, t.x); o.y = __fadd_rn(o.y, t.y); o.z = __fadd_rn(o.z, t.z); o.w = __fadd_rn(o.w, t.w);
                }
                if (owns[q]) {
                    const int c0 = coff[q];
                    if (a.att_src) {
                        const float4 s4 = *reinterpret_cast<const float4*>(a.att_src + c0);
                        const float4 d4 = *reinterpret_cast<const float4*>(a.att_dst + c0);
                        o.x += dsum * s4.x + dd * d4.x; o.y += dsum * s4.y + dd * d4.y;
                        o.z += dsum * s4.z + dd * d4.z; o.w += dsum * s4.w + dd * d4.w;
                    }
                    if (c0 + 1 >= a.C) o.y = 0.f;
                    if (c0 + 2 >= a.C) o.z = 0.f;
                    if (c0 + 3 >= a.C) o.w = 0.f;
                    *reinterpret_cast<float4*>(a.out + row * a.ldo + c0) = o;
                }
            }
            if (lane == 0) a.da[row * a.ld_da] = dsum;
        }
    }
}

bool vec4_ok(const float* p, int64_t ld, int nchunks) { return p && hgv_aligned(p, 16) && ld % 4 == 0 && ld >= (int64_t)nchunks * 4; }

// lanes per row (>= 8: the lanes of a row also hold one edge each, and the backward pass takes its pivot from the first
// L edges of a row) x chunks per lane (<= 4): the fewest lane slots
void pick_lanes(int nchunks, int* ls, int* cpl) {
    int64_t best = -1;
    *ls = 6; *cpl = 4;
    for (int l = 3; l <= 6; ++l) {
        const int Lc = 1 << l;
        int c = (int)hgv_cdiv(nchunks, Lc);
        if (c > 4) continue;
        if (c < 1) c = 1;
        if (best < 0 || (int64_t)Lc * c < best) { best = (int64_t)Lc * c; *ls = l; *cpl = c; }
    }
}

int fill_common(GatArgs& a, const char* who, const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N,
                const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst, float slope, float p,
                const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int C, int* cpl) {
    HGV_REQUIRE(rowptr && col && perm && a_src && a_dst, HGV_EINVAL, "%s: null graph / logit buffer", who);
    HGV_REQUIRE(p >= 0.f && p < 1.f, HGV_EINVAL, "%s: dropout p outside [0,1)", who);
    HGV_REQUIRE(C > 0 && C <= 1024, HGV_EUNSUPPORTED, "%s: feature width %d outside (0, 1024]", who, C);
    a.rowptr = rowptr; a.col = col; a.perm = perm; a.N = N;
    a.a_src = a_src; a.lds_ = ld_src; a.a_dst = a_dst; a.ldd_ = ld_dst;
    a.slope = slope; a.p = p; a.scale = 1.0f / (float)(1.0 - (double)p);
    a.mask = mask; a.mode = p > 0.f ? (mask ? 1 : 2) : 0;
    a.seed = seed; a.offset = offset; a.offset_dev = offset_dev;
    a.C = C; a.nchunks = (C + 3) / 4;
    pick_lanes(a.nchunks, &a.lshift, cpl);
    return HGV_OK;
}


#define HGV_GAT_LAUNCH(KERN, cpl, grid, stream, a)                                                   \
    switch (cpl) {                                                                                   \
        case 1: hipLaunchKernelGGL(KERN<1>, grid, dim3(128), 0, stream, a); break;                   \
        case 2: hipLaunchKernelGGL(KERN<2>, grid, dim3(128), 0, stream, a); break;                   \
        case 3: hipLaunchKernelGGL(KERN<3>, grid, dim3(128), 0, stream, a); break;                   \
        default: hipLaunchKernelGGL(KERN<4>, grid, dim3(128), 0, stream, a); break;                  \
    }

#define HGV_GAT_LAUNCH_LONG(KERN, cpl, grid, stream, a, rows, count, cap)                                              \
    switch (cpl) {                                                                                                     \
        case 1: hipLaunchKernelGGL(KERN<1>, grid, dim3(64 * GAT_LONG_WAVES), 0, stream, a, rows, count, cap); break;     \
        case 2: hipLaunchKernelGGL(KERN<2>, grid, dim3(64 * GAT_LONG_WAVES), 0, stream, a, rows, count, cap); break;     \
        case 3: hipLaunchKernelGGL(KERN<3>, grid, dim3(64 * GAT_LONG_WAVES), 0, stream, a, rows, count, cap); break;     \
        default: hipLaunchKernelGGL(KERN<4>, grid, dim3(64 * GAT_LONG_WAVES), 0, stream, a, rows, count, cap); break;    \
    }

struct LongList { int threshold; const int32_t* rows; const int32_t* count; int cap; int run; };

int check_long(const char* who, const LongList& l) {
    HGV_REQUIRE(l.threshold >= 0 && (l.threshold == 0 || (l.rows && l.count && l.cap > 0)), HGV_EINVAL, "%s: bad long-row list", who);
    return HGV_OK;
}

int gat_fwd_impl(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N, const float* xw,
                 int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst,
                 float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                 const uint64_t* offset_dev, const float* bias, float* out, int64_t ldo, float* row_max,
                 float* row_den, const LongList& lg, hipStream_t stream) {
    HGV_REQUIRE(N >= 0, HGV_EINVAL, "gat_fwd: negative N");
    if (N == 0) return HGV_OK;
    GatArgs a{}; int cpl;
    int rc = fill_common(a, "gat_fwd", rowptr, col, perm, N, a_src, ld_src, a_dst, ld_dst, slope, p, mask, seed, offset,
                         offset_dev, C, &cpl);
    if (rc != HGV_OK) return rc;
    if ((rc = check_long("gat_fwd", lg)) != HGV_OK) return rc;
    HGV_REQUIRE(row_max && row_den, HGV_EINVAL, "gat_fwd: null row statistics buffer");
    HGV_REQUIRE(vec4_ok(xw, ldx, a.nchunks) && vec4_ok(out, ldo, a.nchunks), HGV_EINVAL,
                "gat_fwd: xw / out must be 16-byte aligned with leading dimensions that are multiples of 4 and cover ceil(C/4)*4 columns");
    a.xw = xw; a.ldx = ldx; a.bias = bias; a.out = out; a.ldo = ldo; a.row_max = row_max; a.row_den = row_den;
    a.long_threshold = lg.threshold;
    const dim3 grid((unsigned)hgv_cdiv(N, 2 * (64 >> a.lshift)));
    HGV_GAT_LAUNCH(k_gat_fwd, cpl, grid, stream, a)
    if (lg.threshold > 0 && lg.run) {
        const int cl = (int)hgv_cdiv(a.nchunks, 64);
        const dim3 lgrid((unsigned)(lg.cap < 128 ? lg.cap : 128));
        HGV_GAT_LAUNCH_LONG(k_gat_fwd_long, cl, lgrid, stream, a, lg.rows, lg.count, lg.cap)
    }
    HGV_CHECK_LAUNCH("gat_fwd");
    return HGV_OK;
}

int gat_bwd_impl(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* rowptr_t,
                 const int32_t* col_t, const int32_t* perm_t, int64_t N, const float* g, int64_t ldg,
                 const float* xw, int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst,
                 int64_t ld_dst, float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                 const uint64_t* offset_dev, const float* row_max, const float* row_den, float* S, float* gxw,
                 int64_t ldgx, float* da_src, float* da_dst, int64_t ld_da, const float* att_src,
                 const float* att_dst, const LongList& lg, const LongList& lg_t, hipStream_t stream) {
    HGV_REQUIRE(N >= 0, HGV_EINVAL, "gat_bwd: negative N");
    if (N == 0) return HGV_OK;
    GatArgs a{}; int cpl;
    int rc = fill_common(a, "gat_bwd", rowptr, col, perm, N, a_src, ld_src, a_dst, ld_dst, slope, p, mask, seed, offset,
                         offset_dev, C, &cpl);
    if (rc != HGV_OK) return rc;
    if ((rc = check_long("gat_bwd", lg)) != HGV_OK || (rc = check_long("gat_bwd", lg_t)) != HGV_OK) return rc;
    HGV_REQUIRE(rowptr_t && col_t && perm_t && row_max && row_den && S && da_src && da_dst && ld_da >= 1, HGV_EINVAL, "gat_bwd: null buffer");
    HGV_REQUIRE((att_src == nullptr) == (att_dst == nullptr), HGV_EINVAL, "gat_bwd: att_src and att_dst go together");
    HGV_REQUIRE(!att_src || (hgv_aligned(att_src, 16) && hgv_aligned(att_dst, 16)), HGV_EINVAL,
                "gat_bwd: att_src / att_dst must be 16-byte aligned vectors padded to ceil(C/4)*4 floats");
    HGV_REQUIRE(vec4_ok(xw, ldx, a.nchunks) && vec4_ok(g, ldg, a.nchunks) && vec4_ok(gxw, ldgx, a.nchunks), HGV_EINVAL,
                "gat_bwd: g / xw / gxw must be 16-byte aligned with leading dimensions that are multiples of 4 and cover ceil(C/4)*4 columns");
    a.xw = xw; a.ldx = ldx; a.g = g; a.ldg = ldg;
    a.row_max = const_cast<float*>(row_max); a.row_den = const_cast<float*>(row_den); a.S = S;
    const dim3 grid((unsigned)hgv_cdiv(N, 2 * (64 >> a.lshift)));
    const int cl = (int)hgv_cdiv(a.nchunks, 64);
    a.da = da_dst; a.ld_da = ld_da;
    a.long_threshold = lg.threshold;
    HGV_GAT_LAUNCH(k_gat_bwd_rows, cpl, grid, stream, a)
    if (lg.threshold > 0 && lg.run) {
        const dim3 lgrid((unsigned)(lg.cap < 128 ? lg.cap : 128));
        HGV_GAT_LAUNCH_LONG(k_gat_bwd_rows_long, cl, lgrid, stream, a, lg.rows, lg.count, lg.cap)
    }
    a.rowptr = rowptr_t; a.col = col_t; a.perm = perm_t; a.out = gxw; a.ldo = ldgx; a.da = da_src; a.da_other = da_dst;
    a.att_src = att_src; a.att_dst = att_dst;
    a.long_threshold = lg_t.threshold;
    HGV_GAT_LAUNCH(k_gat_bwd_cols, cpl, grid, stream, a)
    if (lg_t.threshold > 0 && lg_t.run) {
        const dim3 lgrid((unsigned)(lg_t.cap < 128 ? lg_t.cap : 128));
        HGV_GAT_LAUNCH_LONG(k_gat_bwd_cols_long, cl, lgrid, stream, a, lg_t.rows, lg_t.count, lg_t.cap)
    }
    HGV_CHECK_LAUNCH("gat_bwd");
    return HGV_OK;
}

}  // namespace

// out[i,:] = sum_k dropout(softmax_i(leaky_relu(a_src[col_k] + a_dst[i])))_k * xw[col_k,:] (+ bias); row_max / row_den
// [N] receive the row maximum of the logits and the softmax normaliser (sum + 1e-16) for hgv_gat_bwd.
extern "C" int hgv_gat_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N, const float* xw,
                           int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst,
                           float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                           const uint64_t* offset_dev, const float* bias, float* out, int64_t ldo, float* row_max,
                           float* row_den, hipStream_t stream) {
    return gat_fwd_impl(rowptr, col, perm, N, xw, ldx, C, a_src, ld_src, a_dst, ld_dst, slope, p, mask, seed, offset, offset_dev,
                        bias, out, ldo, row_max, row_den, LongList{0, nullptr, nullptr, 0, 0}, stream);
}

// hgv_gat_fwd with the rows longer than long_threshold (> 0; listed in long_rows[min(*long_count, long_cap)] by
// hgv_csr_long_rows with the same threshold) summed by a workgroup each; run_long_pass = 0 skips that pass when the caller
// knows the list is empty.  For those rows the additions are blocked (eight parts in order), not sequential.
extern "C" int hgv_gat_fwd_long(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N, const float* xw,
                                int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst,
                                float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, const float* bias, float* out, int64_t ldo, float* row_max,
                                float* row_den, int long_threshold, const int32_t* long_rows, const int32_t* long_count,
                                int long_cap, int run_long_pass, hipStream_t stream) {
    HGV_REQUIRE(long_threshold > 0, HGV_EINVAL, "gat_fwd_long: threshold must be positive");
    return gat_fwd_impl(rowptr, col, perm, N, xw, ldx, C, a_src, ld_src, a_dst, ld_dst, slope, p, mask, seed, offset, offset_dev,
                        bias, out, ldo, row_max, row_den, LongList{long_threshold, long_rows, long_count, long_cap, run_long_pass}, stream);
}

// Backward of hgv_gat_fwd for the output gradient g [N, C]: gxw [N, C] (d xw), da_src [N], da_dst [N].  The by-target
// CSR (rowptr / col / perm) and the by-source CSR (rowptr_t / col_t / perm_t) of the same edge list, the forward's
// noise arguments and its row_max / row_den are needed; S [2N] is scratch.  da_src / da_dst are written with element stride
// ld_da (two columns of one [N, ld] matrix).  att_src / att_dst (nullable, [ceil(C/4)*4], 16-byte aligned): when the logits
// were formed as a_src = <xw, att_src>, a_dst = <xw, att_dst>, the gradient that reaches xw through them
// (da_src[j] att_src + da_dst[j] att_dst) is added to gxw in the same pass.
extern "C" int hgv_gat_bwd(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* rowptr_t,
                           const int32_t* col_t, const int32_t* perm_t, int64_t N, const float* g, int64_t ldg,
                           const float* xw, int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst,
                           int64_t ld_dst, float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                           const uint64_t* offset_dev, const float* row_max, const float* row_den, float* S, float* gxw,
                           int64_t ldgx, float* da_src, float* da_dst, int64_t ld_da, const float* att_src,
                           const float* att_dst, hipStream_t stream) {
    const LongList none{0, nullptr, nullptr, 0, 0};
    return gat_bwd_impl(rowptr, col, perm, rowptr_t, col_t, perm_t, N, g, ldg, xw, ldx, C, a_src, ld_src, a_dst, ld_dst, slope, p,
                        mask, seed, offset, offset_dev, row_max, row_den, S, gxw, ldgx, da_src, da_dst, ld_da, att_src, att_dst,
                        none, none, stream);
}

// hgv_gat_bwd with hub rows of both CSRs handled by a workgroup each (lists as for hgv_gat_fwd_long: the first for the
// by-target CSR, the second - suffix _t - for the by-source CSR).
extern "C" int hgv_gat_bwd_long(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* rowptr_t,
                                const int32_t* col_t, const int32_t* perm_t, int64_t N, const float* g, int64_t ldg,
                                const float* xw, int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst,
                                int64_t ld_dst, float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, const float* row_max, const float* row_den, float* S, float* gxw,
                                int64_t ldgx, float* da_src, float* da_dst, int64_t ld_da, const float* att_src,
                                const float* att_dst, int long_threshold, const int32_t* long_rows, const int32_t* long_count,
                                int long_cap, int run_long_pass, int long_threshold_t, const int32_t* long_rows_t,
                                const int32_t* long_count_t, int long_cap_t, int run_long_pass_t, hipStream_t stream) {
    HGV_REQUIRE(long_threshold > 0 && long_threshold_t > 0, HGV_EINVAL, "gat_bwd_long: thresholds must be positive");
    return gat_bwd_impl(rowptr, col, perm, rowptr_t, col_t, perm_t, N, g, ldg, xw, ldx, C, a_src, ld_src, a_dst, ld_dst, slope, p,
                        mask, seed, offset, offset_dev, row_max, row_den, S, gxw, ldgx, da_src, da_dst, ld_da, att_src, att_dst,
                        LongList{long_threshold, long_rows, long_count, long_cap, run_long_pass},
                        LongList{long_threshold_t, long_rows_t, long_count_t, long_cap_t, run_long_pass_t}, stream);
}
