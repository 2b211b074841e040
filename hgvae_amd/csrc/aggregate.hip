// Neighbour aggregation over a CSR: out[i,:] = rs_i * sum_{k in row i} w_k * x[col[k],:] (+ add[i,:]).
//
// Replaces PyG's index_select -> scatter_add_ -> count -> divide sequence behind
// SAGEConv / GraphConv / GCNConv / GATConv (reference call sites
// gvae/gvae_skipcon.py:105-118, gvae/gvae_skipcon_attention.py:109,
// autoencoder/vae_with_pool.py:39) and, run on the transposed CSR, its autograd.
//
// Mapping: a group of L lanes (L = power of two <= 64) owns one destination row; lane l
// of the group owns feature chunks l, l+L, ... (VEC floats each, 16-byte loads when the
// layout allows).  The row's segment is walked in order, so the unweighted sum is
// bit-identical to a sequential scatter_add_ in stable edge order.  No atomics, no
// cross-lane reduction: every output element has exactly one writer.
// Column indices are fetched L at a time with one coalesced load and broadcast with
// ds_bpermute (__shfl), so the dependent col -> row-address chain is paid once per L edges.
#include "common.h"

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using T = float; };
template <> struct VecT<2> { using T = float2; };
template <> struct VecT<4> { using T = float4; };

template <int VEC> __device__ __forceinline__ void vzero(typename VecT<VEC>::T& a);
template <> __device__ __forceinline__ void vzero<1>(float& a) { a = 0.f; }
template <> __device__ __forceinline__ void vzero<2>(float2& a) { a = make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ void vzero<4>(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

// acc += v            (unweighted: plain adds, bit-faithful to scatter_add_)
__device__ __forceinline__ void vadd(float& a, const float& v) { a = __fadd_rn(a, v); }
__device__ __forceinline__ void vadd(float2& a, const float2& v) { a.x = __fadd_rn(a.x, v.x); a.y = __fadd_rn(a.y, v.y); }
__device__ __forceinline__ void vadd(float4& a, const float4& v) {
    a.x = __fadd_rn(a.x, v.x); a.y = __fadd_rn(a.y, v.y); a.z = __fadd_rn(a.z, v.z); a.w = __fadd_rn(a.w, v.w);
}
// acc += w * v        (product rounded, then added: the order PyG's message * alpha -> scatter_add_ uses)
__device__ __forceinline__ void vmad(float& a, float w, const float& v) { a = __fadd_rn(a, __fmul_rn(w, v)); }
__device__ __forceinline__ void vmad(float2& a, float w, const float2& v) {
    a.x = __fadd_rn(a.x, __fmul_rn(w, v.x)); a.y = __fadd_rn(a.y, __fmul_rn(w, v.y));
}
__device__ __forceinline__ void vmad(float4& a, float w, const float4& v) {
    a.x = __fadd_rn(a.x, __fmul_rn(w, v.x)); a.y = __fadd_rn(a.y, __fmul_rn(w, v.y));
    a.z = __fadd_rn(a.z, __fmul_rn(w, v.z)); a.w = __fadd_rn(a.w, __fmul_rn(w, v.w));
}
__device__ __forceinline__ void vdiv(float& a, float d) { a = a / d; }
__device__ __forceinline__ void vdiv(float2& a, float d) { a.x /= d; a.y /= d; }
__device__ __forceinline__ void vdiv(float4& a, float d) { a.x /= d; a.y /= d; a.z /= d; a.w /= d; }
__device__ __forceinline__ void vscale(float& a, float s) { a *= s; }
__device__ __forceinline__ void vscale(float2& a, float s) { a.x *= s; a.y *= s; }
__device__ __forceinline__ void vscale(float4& a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }

struct AggArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* x;
    int64_t ldx;
    int64_t n_rows;
    int nchunks;           // ceil(F / VEC)
    int lshift;            // log2(L)
    int row_mode;          // 0: none, 1: divide by max(row length, 1), 2: multiply by row_scale[i]
    const float* row_scale;
    const float* edge_w;   // per CSR position, nullable
    const float* nbr_w;    // indexed by col[k], nullable
    const float* add;      // [n_rows, ldadd], nullable
    int64_t ldadd;
    float* out;
    int64_t ldo;
    int long_threshold;    // > 0: rows with more entries than this are left to k_aggregate_long
    // optional column affine of the SOURCE rows, applied to the finished row instead: with x = a * col_s + col_t column by
    // column, the mean over the neighbours' x rows is (mean over their a rows) * col_s + col_t for a row with neighbours and
    // 0 for a row without (row_mode 1 only); col_s / col_t hold nchunks * VEC floats
    const float* col_s; const float* col_t;
    // optional indicator column: column ind_col (one of the pad columns [F, nchunks * VEC)) of the output row receives
    // 1.0f when the row has neighbours and 0.0f when it has none; -1: pad columns are whatever the sums give
    int ind_col;
};

__device__ __forceinline__ void vset(float& a, int j, float v) { a = v; }
__device__ __forceinline__ void vset(float2& a, int j, float v) { if (j == 0) a.x = v; else a.y = v; }
__device__ __forceinline__ void vset(float4& a, int j, float v) { if (j == 0) a.x = v; else if (j == 1) a.y = v; else if (j == 2) a.z = v; else a.w = v; }

__device__ __forceinline__ void vaffine(float& a, const float* s, const float* t, bool any) { a = __fmaf_rn(a, s[0], any ? t[0] : 0.f); }
__device__ __forceinline__ void vaffine(float2& a, const float* s, const float* t, bool any) {
    a.x = __fmaf_rn(a.x, s[0], any ? t[0] : 0.f); a.y = __fmaf_rn(a.y, s[1], any ? t[1] : 0.f);
}
__device__ __forceinline__ void vaffine(float4& a, const float* s, const float* t, bool any) {
    a.x = __fmaf_rn(a.x, s[0], any ? t[0] : 0.f); a.y = __fmaf_rn(a.y, s[1], any ? t[1] : 0.f);
    a.z = __fmaf_rn(a.z, s[2], any ? t[2] : 0.f); a.w = __fmaf_rn(a.w, s[3], any ? t[3] : 0.f);
}

// NB neighbour rows (x CPL chunks) of one segment position in flight per lane; adds stay in segment order
template <int VEC, bool WEIGHTED, int CPL, int NB>
__device__ __forceinline__ void agg_batch(typename VecT<VEC>::T (&acc)[CPL], const AggArgs& a, const int (&coff)[CPL],
                                          int my_col, float my_w, int j, int L) {
    using V = typename VecT<VEC>::T;
    V v[NB][CPL];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const float* r = a.x + (int64_t)__shfl(my_col, j + u, L) * a.ldx;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[u][q] = *reinterpret_cast<const V*>(r + coff[q]);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const float w = WEIGHTED ? __shfl(my_w, j + u, L) : 1.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            if (WEIGHTED) vmad(acc[q], w, v[u][q]);
            else vadd(acc[q], v[u][q]);
        }
    }
}

// CPL = feature chunks per lane: lane l of a group owns chunks l, l + L, ..., l + (CPL-1) L of a pass.
// Several chunks per lane mean fewer lanes per row (more rows per wave, fewer idle lanes when the
// chunk count is not a power of two: 35 chunks = 16 lanes x 3 instead of 64 x 1) and one column
// index / address computation for CPL row loads.
template <int VEC, bool WEIGHTED, int CPL, int NB>
__global__ __launch_bounds__(128) void k_aggregate(const AggArgs a) {
    using V = typename VecT<VEC>::T;
    const int lane = threadIdx.x & 63;
    const int L = 1 << a.lshift;
    const int lr = lane & (L - 1);
    const int sub = lane >> a.lshift;
    const int rows_per_wave = 64 >> a.lshift;
    const int blk = hgv_xcd_remap(blockIdx.x, gridDim.x);
    const int64_t wave = (int64_t)blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * rows_per_wave + sub;
    if (row >= a.n_rows) return;
    const int start = a.rowptr[row], end = a.rowptr[row + 1];
    if (a.long_threshold > 0 && end - start > a.long_threshold) return;     // the whole group leaves: a long row has its own kernel

    // Every lane of the group walks the segment (it is the shuffle source for its slot of
    // column indices); lanes past the last chunk shadow the last chunk and store nothing.
    for (int c0 = 0; c0 < a.nchunks; c0 += L * CPL) {
        bool owns[CPL];
        int coff[CPL];                      // float offset of the lane's chunks inside a row
        V acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = c0 + q * L + lr;
            owns[q] = c < a.nchunks;
            coff[q] = (owns[q] ? c : a.nchunks - 1) * VEC;
            vzero<VEC>(acc[q]);
        }
        for (int base = start; base < end; base += L) {
            const int k = base + lr;
            int my_col = 0;
            float my_w = 1.f;
            if (k < end) {
                my_col = a.col[k];
                if (WEIGHTED) {
                    if (a.edge_w) my_w = a.edge_w[k];
                    if (a.nbr_w) my_w *= a.nbr_w[my_col];
                }
            }
            const int cnt = min(L, end - base);
            int j = 0;
            for (; j + NB <= cnt; j += NB) agg_batch<VEC, WEIGHTED, CPL, NB>(acc, a, coff, my_col, my_w, j, L);
            if (NB > 2)
                for (; j + 2 <= cnt; j += 2) agg_batch<VEC, WEIGHTED, CPL, 2>(acc, a, coff, my_col, my_w, j, L);
            for (; j < cnt; ++j) agg_batch<VEC, WEIGHTED, CPL, 1>(acc, a, coff, my_col, my_w, j, L);
        }
        const int d = end - start;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            if (a.row_mode == 1) vdiv(acc[q], (float)(d > 1 ? d : 1));
            else if (a.row_mode == 2) vscale(acc[q], a.row_scale[row]);
            if (a.col_s) vaffine(acc[q], a.col_s + coff[q], a.col_t + coff[q], d > 0);
            if (a.ind_col >= coff[q] && a.ind_col < coff[q] + VEC) vset(acc[q], a.ind_col - coff[q], d > 0 ? 1.f : 0.f);
            if (owns[q]) {
                if (a.add) vadd(acc[q], *reinterpret_cast<const V*>(a.add + row * a.ldadd + coff[q]));
                *reinterpret_cast<V*>(a.out + row * a.ldo + coff[q]) = acc[q];
            }
        }
    }
}

// Rows far longer than the average ("hubs").  A lane group walks its row alone, a few neighbour rows at a time, so one
// row of 2 000 entries takes longer than the rest of a 100 k-row launch together.  Such rows are listed once per CSR
// (k_long_rows) and summed by a whole 1024-thread workgroup each: wave w of 16 takes the w-th sixteenth of the segment
// (in order), a lane owns one feature chunk per pass, and the sixteen partial sums are added in wave order through LDS
// - no atomics, the same result on every run; for these rows the order of the additions is blocked, not sequential.
constexpr int LONG_WAVES = 16;

__global__ __launch_bounds__(256) void k_long_rows(const int32_t* __restrict__ rowptr, int64_t n_rows, int threshold,
                                                   int32_t* __restrict__ rows, int cap, int32_t* __restrict__ count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
        if (rowptr[i + 1] - rowptr[i] > threshold) {
            const int pos = atomicAdd(count, 1);          // list order is irrelevant: every listed row is summed on its own
            if (pos < cap) rows[pos] = (int32_t)i;
        }
    }
}

template <int VEC, bool WEIGHTED>
__global__ __launch_bounds__(64 * LONG_WAVES) void k_aggregate_long(const AggArgs a, const int32_t* __restrict__ rows,
                                                                    const int32_t* __restrict__ count, int cap) {
    using V = typename VecT<VEC>::T;
    __shared__ float4 sh4[LONG_WAVES * 64];
    V* sh = reinterpret_cast<V*>(sh4);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_long = min(count[0], cap);
    for (int idx = blockIdx.x; idx < n_long; idx += gridDim.x) {
        const int64_t row = rows[idx];
        const int start = a.rowptr[row], end = a.rowptr[row + 1];
        const int len = end - start;
        const int s0 = start + (int)((int64_t)len * w / LONG_WAVES), s1 = start + (int)((int64_t)len * (w + 1) / LONG_WAVES);
        for (int c0 = 0; c0 < a.nchunks; c0 += 64) {
            const int c = c0 + lane;
            const bool owns = c < a.nchunks;
            const int coff = (owns ? c : a.nchunks - 1) * VEC;
            V acc;
            vzero<VEC>(acc);
            for (int base = s0; base < s1; base += 64) {
                const int k = base + lane;
                int my_col = 0;
                float my_w = 1.f;
                if (k < s1) {
                    my_col = a.col[k];
                    if (WEIGHTED) {
                        if (a.edge_w) my_w = a.edge_w[k];
                        if (a.nbr_w) my_w *= a.nbr_w[my_col];
                    }
                }
                const int cnt = min(64, s1 - base);
                int j = 0;
                for (; j + 8 <= cnt; j += 8) {          // eight neighbour rows in flight per wave: the row is all this workgroup has
                    V v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const V*>(a.x + (int64_t)__shfl(my_col, j + u, 64) * a.ldx + coff);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (WEIGHTED) vmad(acc, __shfl(my_w, j + u, 64), v[u]);
                        else vadd(acc, v[u]);
                    }
                }
                for (; j < cnt; ++j) {
                    const V v = *reinterpret_cast<const V*>(a.x + (int64_t)__shfl(my_col, j, 64) * a.ldx + coff);
                    if (WEIGHTED) vmad(acc, __shfl(my_w, j, 64), v);
                    else vadd(acc, v);
                }
            }
            sh[w * 64 + lane] = acc;
            __syncthreads();
            if (w == 0) {
                V tot = sh[lane];
                for (int q = 1; q < LONG_WAVES; ++q) vadd(tot, sh[q * 64 + lane]);
                if (a.row_mode == 1) vdiv(tot, (float)(len > 1 ? len : 1));
                else if (a.row_mode == 2) vscale(tot, a.row_scale[row]);
                if (a.col_s) vaffine(tot, a.col_s + coff, a.col_t + coff, len > 0);
                if (a.ind_col >= coff && a.ind_col < coff + VEC) vset(tot, a.ind_col - coff, len > 0 ? 1.f : 0.f);
                if (owns) {
                    if (a.add) vadd(tot, *reinterpret_cast<const V*>(a.add + row * a.ldadd + coff));
                    *reinterpret_cast<V*>(a.out + row * a.ldo + coff) = tot;
                }
            }
            __syncthreads();
        }
    }
}

bool vec_ok(int vec, int F, const float* p, int64_t ld) {
    if (p == nullptr) return true;
    const int64_t need = hgv_cdiv(F, vec) * vec;
    return hgv_aligned(p, 4 * vec) && (ld % vec == 0) && ld >= need;
}

int launch_aggregate(AggArgs a, int F, bool weighted, hipStream_t st, const int32_t* long_rows = nullptr,
                     const int32_t* long_count = nullptr, int long_cap = 0) {
    int vec = 4;
    while (vec > 1 && !(vec_ok(vec, F, a.x, a.ldx) && vec_ok(vec, F, a.out, a.ldo) && vec_ok(vec, F, a.add, a.ldadd)))
        vec >>= 1;
    a.nchunks = (int)hgv_cdiv(F, vec);
    // lanes per row L (power of two, >= 8 so that a group reads >= 128 contiguous bytes with 16-byte
    // chunks) and chunks per lane (<= 4): the fewest lane slots L x CPL >= nchunks; ties go to the
    // narrower group (more rows per wave, fewer index shuffles per row).  (8 lanes x 5 chunks for the 35
    // chunks of width 138 - 40 slots instead of 48 - is 3 % faster for the plain sum in isolation, equal
    // inside the train step and 5 % slower for the weighted sum: not used.)
    int ls = 0, cpl = 1;
    {
        int64_t best = -1;
        for (int l = 0; l <= 6; ++l) {
            const int Lc = 1 << l;
            if (Lc < 8 && Lc < a.nchunks) continue;
            int c = (int)hgv_cdiv(a.nchunks, Lc);
            if (c > 4) continue;
            if (c < 1) c = 1;
            const int64_t slots = (int64_t)Lc * c;
            if (best < 0 || slots < best) { best = slots; ls = l; cpl = c; }
        }
        if (best < 0) {                    // more than 256 chunks: 64 lanes, several passes
            ls = 6; cpl = 4;
        }
    }
    // two waves per workgroup: measured 3 % faster than four (finer dispatch), larger workgroups are slower
    const int wpb = 2;
    a.lshift = ls;
    const int rows_per_block = wpb * (64 >> ls);
    const int64_t nblocks = hgv_cdiv(a.n_rows, rows_per_block);
    if (nblocks == 0) return HGV_OK;
    HGV_REQUIRE(nblocks < ((int64_t)1 << 31), HGV_EUNSUPPORTED, "aggregate: too many rows");
    dim3 g((unsigned)nblocks), b(64 * wpb);
    // neighbour rows in flight per lane: 4 x (1-2 chunks) or 2 x (3-4 chunks) - measured: occupancy beats deeper batches
#define HGV_AGG_CPL(V, W)                                                                          \
    switch (cpl) {                                                                                 \
        case 1: hipLaunchKernelGGL((k_aggregate<V, W, 1, 4>), g, b, 0, st, a); break;              \
        case 2: hipLaunchKernelGGL((k_aggregate<V, W, 2, 4>), g, b, 0, st, a); break;              \
        case 3: hipLaunchKernelGGL((k_aggregate<V, W, 3, 2>), g, b, 0, st, a); break;              \
        default: hipLaunchKernelGGL((k_aggregate<V, W, 4, 2>), g, b, 0, st, a); break;             \
    }
#define HGV_AGG_LAUNCH(V)                                                                          \
    if (weighted) { HGV_AGG_CPL(V, true) } else { HGV_AGG_CPL(V, false) }
    if (vec == 4) { HGV_AGG_LAUNCH(4) } else if (vec == 2) { HGV_AGG_LAUNCH(2) } else { HGV_AGG_LAUNCH(1) }
#undef HGV_AGG_LAUNCH
#undef HGV_AGG_CPL
    if (long_rows != nullptr && long_cap > 0) {
        dim3 lg((unsigned)(long_cap < 128 ? long_cap : 128)), lb(64 * LONG_WAVES);
#define HGV_AGG_LONG(V)                                                                                              \
        if (weighted) hipLaunchKernelGGL((k_aggregate_long<V, true>), lg, lb, 0, st, a, long_rows, long_count, long_cap);   \
        else hipLaunchKernelGGL((k_aggregate_long<V, false>), lg, lb, 0, st, a, long_rows, long_count, long_cap);
        if (vec == 4) { HGV_AGG_LONG(4) } else if (vec == 2) { HGV_AGG_LONG(2) } else { HGV_AGG_LONG(1) }
#undef HGV_AGG_LONG
    }
    HGV_CHECK_LAUNCH("aggregate");
    return HGV_OK;
}

}  // namespace

// General entry point.  row_mode: 0 = plain sum, 1 = mean over the row (divide by
// max(row length, 1)), 2 = multiply by row_scale[i].
extern "C" int hgv_agg_fwd(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* x, int64_t ldx,
                           int F, int row_mode, const float* row_scale, const float* edge_w, const float* nbr_w,
                           const float* add, int64_t ldadd, float* out, int64_t ldo, hipStream_t stream) {
    HGV_REQUIRE(n_rows >= 0 && F >= 0, HGV_EINVAL, "agg_fwd: negative size");
    if (n_rows == 0 || F == 0) return HGV_OK;
    // col may be null for a graph without edges (an empty tensor has no storage): no row then dereferences it
    HGV_REQUIRE(rowptr && x && out, HGV_EINVAL, "agg_fwd: null buffer");
    HGV_REQUIRE(ldx >= F && ldo >= F && (!add || ldadd >= F), HGV_EINVAL, "agg_fwd: leading dimension < F");
    HGV_REQUIRE(row_mode >= 0 && row_mode <= 2 && (row_mode != 2 || row_scale), HGV_EINVAL, "agg_fwd: bad row_mode");
    AggArgs a{rowptr, col, x, ldx, n_rows, 0, 0, row_mode, row_scale, edge_w, nbr_w, add, ldadd, out, ldo, 0, nullptr, nullptr, -1};
    return launch_aggregate(a, F, edge_w != nullptr || nbr_w != nullptr, stream);
}

// Rows of a CSR with more than `threshold` entries, in no particular order: rows[0 .. min(count, cap)) and their number in
// count[0], which must be zero on entry.  cap >= E / threshold + 1 always suffices.
extern "C" int hgv_csr_long_rows(const int32_t* rowptr, int64_t n_rows, int threshold, int32_t* rows, int cap,
                                 int32_t* count, hipStream_t stream) {
    HGV_REQUIRE(n_rows >= 0 && threshold > 0 && cap > 0 && rows && count, HGV_EINVAL, "csr_long_rows: bad argument");
    if (n_rows == 0) return HGV_OK;
    HGV_REQUIRE(rowptr, HGV_EINVAL, "csr_long_rows: null rowptr");
    const int64_t blocks = hgv_cdiv(n_rows, 256);
    hipLaunchKernelGGL(k_long_rows, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, rowptr, n_rows, threshold, rows, cap, count);
    HGV_CHECK_LAUNCH("csr_long_rows");
    return HGV_OK;
}

// hgv_agg_fwd for graphs with hubs: rows longer than `threshold` are skipped by the row-per-lane-group kernel; with
// run_long_pass != 0 a second launch sums the rows listed by hgv_csr_long_rows (same threshold) with one workgroup per
// row.  run_long_pass = 0 is for callers that know the list is empty.  Every row is written by exactly one of the two.
static int agg_fwd_long_any(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* x, int64_t ldx,
                                int F, int row_mode, const float* row_scale, const float* edge_w, const float* nbr_w,
                                const float* add, int64_t ldadd, float* out, int64_t ldo, int threshold,
                                const int32_t* long_rows, const int32_t* long_count, int long_cap, int run_long_pass,
                                const float* col_s, const float* col_t, hipStream_t stream, int ind_col = -1) {
    HGV_REQUIRE(n_rows >= 0 && F >= 0, HGV_EINVAL, "agg_fwd_long: negative size");
    if (n_rows == 0 || F == 0) return HGV_OK;
    HGV_REQUIRE(rowptr && x && out, HGV_EINVAL, "agg_fwd_long: null buffer");
    HGV_REQUIRE(ldx >= F && ldo >= F && (!add || ldadd >= F), HGV_EINVAL, "agg_fwd_long: leading dimension < F");
    HGV_REQUIRE(row_mode >= 0 && row_mode <= 2 && (row_mode != 2 || row_scale), HGV_EINVAL, "agg_fwd_long: bad row_mode");
    HGV_REQUIRE(threshold > 0 && long_rows && long_count && long_cap > 0, HGV_EINVAL, "agg_fwd_long: bad long-row list");
    HGV_REQUIRE((col_s == nullptr) == (col_t == nullptr) && (!col_s || (row_mode == 1 && !add && hgv_aligned(col_s, 16) && hgv_aligned(col_t, 16))),
                HGV_EINVAL, "agg_fwd_affine: column scale and shift come together, 16-byte aligned, with the row mean and no addend");
    HGV_REQUIRE(ind_col < 0 || (ind_col >= F && ind_col < (F + 3) / 4 * 4 && ldo > ind_col && hgv_aligned(x, 16) && hgv_aligned(out, 16) &&
                                ldx % 4 == 0 && ldo % 4 == 0), HGV_EINVAL, "agg_fwd_ind: the indicator column must be a pad column of 16-byte rows");
    AggArgs a{rowptr, col, x, ldx, n_rows, 0, 0, row_mode, row_scale, edge_w, nbr_w, add, ldadd, out, ldo, threshold, col_s, col_t, ind_col};
    return launch_aggregate(a, F, edge_w != nullptr || nbr_w != nullptr, stream, run_long_pass ? long_rows : nullptr, long_count,
                            long_cap);
}

extern "C" int hgv_agg_fwd_long(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* x, int64_t ldx,
                                int F, int row_mode, const float* row_scale, const float* edge_w, const float* nbr_w,
                                const float* add, int64_t ldadd, float* out, int64_t ldo, int threshold,
                                const int32_t* long_rows, const int32_t* long_count, int long_cap, int run_long_pass,
                                hipStream_t stream) {
    return agg_fwd_long_any(rowptr, col, n_rows, x, ldx, F, row_mode, row_scale, edge_w, nbr_w, add, ldadd, out, ldo, threshold,
                            long_rows, long_count, long_cap, run_long_pass, nullptr, nullptr, stream);
}

// hgv_agg_fwd_long (row_mode 1, no addend) over source rows that are handed over un-normalised: x = a * col_s + col_t
// column by column (col_s / col_t: 4 * ceil(F / 4) floats, 16-byte aligned).  out_i = mean_j(a_j) * col_s + col_t for a row
// with neighbours, 0 otherwise - the mean aggregation of x without x ever being written.
extern "C" int hgv_agg_fwd_affine(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* a, int64_t lda, int F,
                                  const float* col_s, const float* col_t, float* out, int64_t ldo, int threshold,
                                  const int32_t* long_rows, const int32_t* long_count, int long_cap, int run_long_pass,
                                  hipStream_t stream) {
    HGV_REQUIRE(col_s && col_t, HGV_EINVAL, "agg_fwd_affine: null column scale / shift");
    return agg_fwd_long_any(rowptr, col, n_rows, a, lda, F, 1, nullptr, nullptr, nullptr, nullptr, 0, out, ldo, threshold,
                            long_rows, long_count, long_cap, run_long_pass, col_s, col_t, stream);
}

// Backward of the mean/sum aggregation with respect to x: run over the transposed CSR
// (rows = source nodes, col_t = destination of each out-edge):
//   gx[j,:] = sum_{k in rowT j} inv_deg[col_t[k]] * g[col_t[k],:]  (+ add[j,:])
// inv_deg = 1/max(in-degree, 1) for mean aggregation, null for sum aggregation.
extern "C" int hgv_agg_bwd(const int32_t* rowptr_t, const int32_t* col_t, int64_t n_rows, const float* g,
                           int64_t ldg, int F, const float* inv_deg, const float* edge_w_t, const float* add,
                           int64_t ldadd, float* gx, int64_t ldgx, hipStream_t stream) {
    return hgv_agg_fwd(rowptr_t, col_t, n_rows, g, ldg, F, 0, nullptr, edge_w_t, inv_deg, add, ldadd, gx, ldgx,
                       stream);
}

// Mean aggregation (hgv_agg_fwd_long with row_mode 1, no weights, no addend) that also writes an indicator into the pad
// column ind_col (F <= ind_col < 4 * ceil(F / 4)) of every output row: 1.0f for a row with neighbours, 0.0f for one
// without.  For source rows that stand for x = d * s + t: mean_j x_j = mean_j(d_j) * s + t * indicator, so a projection
// that follows can take [mean(d) | indicator] as its operand with weights [W diag(s) | W t] (hgv_pack16h_weights_affine)
// and the aggregation itself costs what it costs on an ordinary matrix.
extern "C" int hgv_agg_fwd_ind(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* d, int64_t ldd, int F,
                               float* out, int64_t ldo, int ind_col, int threshold, const int32_t* long_rows,
                               const int32_t* long_count, int long_cap, int run_long_pass, hipStream_t stream) {
    HGV_REQUIRE(ind_col >= 0, HGV_EINVAL, "agg_fwd_ind: no indicator column");
    return agg_fwd_long_any(rowptr, col, n_rows, d, ldd, F, 1, nullptr, nullptr, nullptr, nullptr, 0, out, ldo, threshold,
                            long_rows, long_count, long_cap, run_long_pass, nullptr, nullptr, stream, ind_col);
}
