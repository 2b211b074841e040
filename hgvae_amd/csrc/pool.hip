// SAGPooling support kernels: per-graph top-k selection, adjacency filtering, row gathers.
//
// Reference: SAGPooling(hidden) at autoencoder/vae_with_pool.py:25,36,39 (PyG defaults
// ratio=0.5, GraphConv scorer, tanh) and the GCNConv-scored variant at
// autoencoder/vae_custom_pool.py:56-58; unpooling gathers x[perm], batch[perm],
// perm[edge_index_cg] at autoencoder/vae_with_pool.py:42-50.
//
//   perm      : per graph (graphs in order) the ceil(ratio * n_g) nodes of highest score,
//               descending score, ties by ascending node index
//   filter_adj: keep edges whose two ends are kept, relabel to positions in perm, edge order kept
// All index results are exact (integer sorts/scans); nothing here uses atomics.
#include "common.h"

namespace {

// Monotone map float -> uint32 such that a DESCENDING float order is an ASCENDING key order.
// -0.0 is folded onto +0.0 first (torch.sort treats them as equal).
__global__ void k_topk_keys(const float* __restrict__ score, int64_t n, uint32_t* __restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = score[i] + 0.0f;
        uint32_t u = __float_as_uint(s);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending in s
        keys[i] = ~u;                                     // descending in s
    }
}

__global__ void k_gather_batch_keys(const int64_t* __restrict__ batch, const int32_t* __restrict__ ids, int64_t n,
                                    uint32_t* __restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        keys[i] = (uint32_t)batch[ids[i]];
}

// ids_sorted: node ids ordered by (graph, score desc, id asc); position p belongs to graph
// g = batch[ids_sorted[p]] with rank r = p - ptr[g]; kept when r < k[g].
__global__ void k_topk_select(const int32_t* __restrict__ ids_sorted, const int64_t* __restrict__ batch,
                              const int64_t* __restrict__ ptr, const int64_t* __restrict__ k,
                              const int64_t* __restrict__ kptr, int64_t n, int64_t* __restrict__ perm) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t node = ids_sorted[p];
        const int64_t g = batch[node];
        const int64_t r = p - ptr[g];
        if (r < k[g]) perm[kptr[g] + r] = node;
    }
}

// One workgroup per graph whose node count fits the LDS table: the graph's (descending-score key, local node index)
// pairs are sorted as 64-bit words by a bitonic network in LDS (total order: ties in the score fall back to the
// ascending node index, the order of a stable sort) and the first k[g] entries go straight to perm.
// The network's steps are taken up to four at a time: a thread loads the 16 words whose indices differ in the four
// stride bits of the group, runs the 32 compare-exchanges in registers and writes the words back, so the 105 steps of
// a 16384-word table cost 32 passes over LDS (the sort is bound by LDS bandwidth: one CU moves the whole table per pass).
// Word i lives at slot i + i / 16, which spreads the 16-word strides of the low groups over the banks.
constexpr int SEG_CAP = 16384;             // words per table
constexpr int SEG_THREADS = 1024;
__host__ __device__ inline int seg_slot(int i) { return i + (i >> 4); }
inline size_t seg_lds_bytes(int m) { return (size_t)(seg_slot(m - 1) + 1) * 8; }

// steps with stride bits b, b-1, ..., b-NST+1 of the merge of width `size`
template <int NST>
__device__ __forceinline__ void topk_group(unsigned long long* tab, int m, int size, int b) {
    constexpr int E = 1 << NST;
    const int lowbit = b - NST + 1;
    for (int t = threadIdx.x; t < (m >> NST); t += SEG_THREADS) {
        const int low = t & ((1 << lowbit) - 1), high = t >> lowbit;
        const int base = (high << (b + 1)) | low;
        const bool up = (base & size) == 0;
        unsigned long long r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = tab[seg_slot(base | (e << lowbit))];
#pragma unroll
        for (int j = NST - 1; j >= 0; --j) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e >> j) & 1) continue;
                const unsigned long long x = r[e], y = r[e | (1 << j)];
                const bool sw = (x > y) == up;
                r[e] = sw ? y : x;
                r[e | (1 << j)] = sw ? x : y;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) tab[seg_slot(base | (e << lowbit))] = r[e];
    }
}

__global__ __launch_bounds__(SEG_THREADS) void k_topk_segment(const float* __restrict__ score, const int64_t* __restrict__ ptr,
                                                              const int64_t* __restrict__ k, const int64_t* __restrict__ kptr,
                                                              int64_t* __restrict__ perm, const int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char seg_raw[];
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(seg_raw);
    const int g = blockIdx.x;
    const int64_t lo = ptr[g];
    const int n = (int)(ptr[g + 1] - lo);
    const int kg = (int)k[g];
    if (n <= 0 || kg <= 0) return;
    if (n > cap) {
        // the caller's max_nodes was too small for this graph: its table would not fit the LDS that was asked for.
        // Stay inside memory and inside the graph (valid node ids, arbitrary choice); the contract names this case.
        const int64_t base = kptr[g];
        for (int r = threadIdx.x; r < kg; r += SEG_THREADS) perm[base + r] = lo + (r < n ? r : n - 1);
        return;
    }
    int m = 1, logm = 0;
    while (m < n) { m <<= 1; ++logm; }     // table size: next power of two, padded with keys that sort last
    for (int i = threadIdx.x; i < m; i += SEG_THREADS) {
        unsigned long long w = ~0ull;
        if (i < n) {
            const float sc = score[lo + i] + 0.0f;                       // -0.0 -> +0.0: torch.sort treats them as equal
            uint32_t u = __float_as_uint(sc);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // ascending in the score
            w = ((unsigned long long)(~u) << 32) | (unsigned)i;           // descending score, then ascending index
        }
        tab[seg_slot(i)] = w;
    }
    __syncthreads();
    for (int ls = 1; ls <= logm; ++ls) {                                   // merge width 2^ls: stride bits ls-1 ... 0
        const int size = 1 << ls;
        for (int b = ls - 1; b >= 0;) {
            const int nst = b + 1 < 4 ? b + 1 : 4;
            if (nst == 4) topk_group<4>(tab, m, size, b);
            else if (nst == 3) topk_group<3>(tab, m, size, b);
            else if (nst == 2) topk_group<2>(tab, m, size, b);
            else topk_group<1>(tab, m, size, b);
            b -= nst;
            __syncthreads();
        }
    }
    const int64_t base = kptr[g];
    for (int r = threadIdx.x; r < kg; r += SEG_THREADS) perm[base + r] = lo + (int64_t)(tab[seg_slot(r)] & 0xffffffffull);
}

__global__ void k_fill_i32(int32_t* __restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_mark_perm(const int64_t* __restrict__ perm, int64_t K, int32_t* __restrict__ map) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < K; q += (int64_t)gridDim.x * blockDim.x)
        map[perm[q]] = (int32_t)q;
}
__global__ void k_edge_flags(const int64_t* __restrict__ ei, int64_t ld_row, int64_t ld_col, int64_t E,
                             const int32_t* __restrict__ map, int32_t* __restrict__ flags) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x)
        flags[e] = (map[ei[e * ld_col]] >= 0 && map[ei[ld_row + e * ld_col]] >= 0) ? 1 : 0;
}
__global__ void k_edge_compact(const int64_t* __restrict__ ei, int64_t ld_row, int64_t ld_col, int64_t E,
                               const int32_t* __restrict__ map, const int32_t* __restrict__ flags,
                               const int32_t* __restrict__ pos, int64_t* __restrict__ out, int64_t ld_out,
                               int32_t* __restrict__ kept_ids, int32_t* __restrict__ count, int fill_tail) {
    const int32_t total = pos[E - 1] + flags[E - 1];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        if (flags[e]) {
            const int32_t q = pos[e];
            out[q] = map[ei[e * ld_col]];
            out[ld_out + q] = map[ei[ld_row + e * ld_col]];
            if (kept_ids) kept_ids[q] = (int32_t)e;
        } else if (fill_tail) {
            // the dropped edges number E - total: dropped edge e is the (e - pos[e])-th of them and zeroes that tail slot,
            // so a caller that works on the whole capacity without reading count back sees valid (zero) indices
            const int64_t q = (int64_t)total + (e - pos[e]);
            out[q] = 0;
            out[ld_out + q] = 0;
            if (kept_ids) kept_ids[q] = 0;
        }
        if (e == E - 1) count[0] = total;
    }
}

// out[q, :] = x[idx[q], :] * (scale ? scale[idx[q]] : 1) with 16-byte accesses: a group of L lanes owns a gathered row,
// lane l moves the 4-float chunks l, l + L, ...; pad columns of the output are written as zeros.
__global__ __launch_bounds__(256) void k_gather_rows4(const float* __restrict__ x, int64_t ldx,
                                                      const int64_t* __restrict__ idx, const float* __restrict__ scale,
                                                      int64_t K, int F, int lshift, float* __restrict__ out, int64_t ldo) {
    const int L = 1 << lshift, lr = threadIdx.x & (L - 1);
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lshift;
    if (q >= K) return;
    const int64_t r = idx[q];
    const float s = scale ? scale[r] : 1.f;
    const int nchunks = (F + 3) >> 2;
    for (int c = lr; c < nchunks; c += L) {
        float4 v = *reinterpret_cast<const float4*>(x + r * ldx + 4 * c);
        if (scale) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
        if (4 * c + 1 >= F) v.y = 0.f;
        if (4 * c + 2 >= F) v.z = 0.f;
        if (4 * c + 3 >= F) v.w = 0.f;
        *reinterpret_cast<float4*>(out + q * ldo + 4 * c) = v;
    }
}

// scalar variant for layouts without 16-byte aligned rows
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ x, int64_t ldx,
                                                     const int64_t* __restrict__ idx, const float* __restrict__ scale,
                                                     int64_t K, int F, float* __restrict__ out, int64_t ldo) {
    const int64_t total = K * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / F;
        const int c = (int)(i - q * F);
        const int64_t r = idx[q];
        const float v = x[r * ldx + c];
        out[q * ldo + c] = scale ? v * scale[r] : v;
    }
}

// backward of out = x[idx] * scale[idx]: gx[idx[q], :] = g[q, :] * scale[idx[q]] (rows not in idx
// must be zeroed by the caller), gscale[idx[q]] = <g[q, :], x[idx[q], :]>.  idx holds distinct rows.
__global__ __launch_bounds__(256) void k_gather_rows_bwd(const float* __restrict__ g, int64_t ldg,
                                                         const float* __restrict__ x, int64_t ldx,
                                                         const int64_t* __restrict__ idx,
                                                         const float* __restrict__ scale, int64_t K, int F,
                                                         float* __restrict__ gx, int64_t ldgx,
                                                         float* __restrict__ gscale) {
    const int lane = threadIdx.x & 63;
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // one wave per gathered row
    if (q >= K) return;
    const int64_t r = idx[q];
    const float s = scale ? scale[r] : 1.f;
    float acc = 0.f;
    for (int c = lane; c < F; c += 64) {
        const float gv = g[q * ldg + c];
        if (gx) gx[r * ldgx + c] = gv * s;
        if (gscale) acc += gv * x[r * ldx + c];
    }
    if (gscale) {
        acc = hgv_wave_sum(acc);
        if (lane == 0) gscale[r] = acc;
    }
}

__global__ void k_gather_i64(const int64_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n,
                             int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = src[idx[i]];
}

int grid1d(int64_t n) {
    int64_t g = hgv_cdiv(n, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int hgv_sort_pairs_u32(const uint32_t*, const int32_t*, int64_t, int, int, uint32_t*, int32_t*, void*, size_t, hipStream_t);
extern "C" size_t hgv_sort_pairs_ws(int64_t);
extern "C" int hgv_exclusive_scan_i32(const int32_t*, int32_t*, int64_t, void*, size_t, hipStream_t);
extern "C" size_t hgv_exclusive_scan_ws(int64_t);

extern "C" size_t hgv_topk_ws(int64_t N) {
    if (N < 0) return 0;
    return hgv_sort_pairs_ws(N) + 4 * hgv_align256((size_t)N * 4);
}

// Per-graph top-k.  batch [N] int64 (graph id per node, ascending), ptr [B+1], k [B], kptr [B+1]
// (exclusive cumsum of k) are device arrays prepared by the caller; perm gets kptr[B] entries.
extern "C" int hgv_topk_perm(const float* score, const int64_t* batch, const int64_t* ptr, const int64_t* k,
                             const int64_t* kptr, int64_t N, int64_t B, int64_t max_nodes, int64_t* perm, void* ws,
                             size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(N >= 0 && B >= 0, HGV_EINVAL, "topk_perm: negative size");
    if (N == 0) return HGV_OK;
    HGV_REQUIRE(score && batch && ptr && k && kptr && perm, HGV_EINVAL, "topk_perm: null buffer");
    if (max_nodes > 0 && max_nodes <= SEG_CAP && B <= 65535 * 16) {
        // every graph fits one workgroup's LDS table: one launch, no global sort passes
        int m = 1;
        while (m < max_nodes) m <<= 1;
        static bool raised[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!raised[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_topk_segment),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_bytes(SEG_CAP));
            HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "topk_perm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            raised[dev] = true;
        }
        hipLaunchKernelGGL(k_topk_segment, dim3((unsigned)B), dim3(SEG_THREADS), seg_lds_bytes(m), stream, score, ptr, k, kptr, perm, m);
        HGV_CHECK_LAUNCH("topk_perm(segmented)");
        return HGV_OK;
    }
    HGV_REQUIRE(ws != nullptr, HGV_EINVAL, "topk_perm: null workspace");
    HGV_REQUIRE(ws_bytes >= hgv_topk_ws(N) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "topk_perm: workspace too small or misaligned");
    char* p = (char*)ws;
    const size_t seg = hgv_align256((size_t)N * 4);
    uint32_t* keys = (uint32_t*)p;
    uint32_t* keys2 = (uint32_t*)(p + seg);
    int32_t* ids = (int32_t*)(p + 2 * seg);
    int32_t* ids2 = (int32_t*)(p + 3 * seg);
    void* sws = p + 4 * seg;
    const size_t sws_bytes = ws_bytes - 4 * seg;
    const int g = grid1d(N);
    hipLaunchKernelGGL(k_topk_keys, dim3(g), dim3(256), 0, stream, score, N, keys);
    int rc = hgv_sort_pairs_u32(keys, nullptr, N, 0, 32, nullptr, ids, sws, sws_bytes, stream);
    if (rc != HGV_OK) return rc;
    int bits = 1;
    while (((int64_t)1 << bits) < B) ++bits;
    hipLaunchKernelGGL(k_gather_batch_keys, dim3(g), dim3(256), 0, stream, batch, ids, N, keys2);
    rc = hgv_sort_pairs_u32(keys2, ids, N, 0, bits, nullptr, ids2, sws, sws_bytes, stream);
    if (rc != HGV_OK) return rc;
    hipLaunchKernelGGL(k_topk_select, dim3(g), dim3(256), 0, stream, ids2, batch, ptr, k, kptr, N, perm);
    HGV_CHECK_LAUNCH("topk_perm");
    return HGV_OK;
}

extern "C" size_t hgv_filter_adj_ws(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return hgv_align256((size_t)N * 4) + 2 * hgv_align256((size_t)E * 4) + hgv_exclusive_scan_ws(E);
}

// filter_adj: edges (any strides) whose two ends are in perm, relabelled to positions in perm, order
// preserved.  out_ei [2, cap] int64 with row stride ld_out (cap >= E is always enough); kept_ids
// (nullable, [cap]) receives the original edge id of every kept edge (to gather edge_attr);
// count[0] receives the number of kept edges.  fill_tail != 0 additionally writes zeros to slots [count, E) of out_ei /
// kept_ids (cap >= E then), for callers that keep working on the whole capacity without reading count back.
extern "C" int hgv_filter_adj(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t N,
                              const int64_t* perm, int64_t K, int64_t* out_ei, int64_t ld_out, int32_t* kept_ids,
                              int32_t* count, int fill_tail, void* ws, size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(E >= 0 && N >= 0 && K >= 0, HGV_EINVAL, "filter_adj: negative size");
    HGV_REQUIRE(count != nullptr, HGV_EINVAL, "filter_adj: count is null");
    if (E == 0) {
        hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, stream, count, 1, 0);
        HGV_CHECK_LAUNCH("filter_adj(empty)");
        return HGV_OK;
    }
    HGV_REQUIRE(edge_index && out_ei && ws && (K == 0 || perm), HGV_EINVAL, "filter_adj: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_filter_adj_ws(E, N) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "filter_adj: workspace too small or misaligned");
    char* p = (char*)ws;
    int32_t* map = (int32_t*)p;                p += hgv_align256((size_t)N * 4);
    int32_t* flags = (int32_t*)p;              p += hgv_align256((size_t)E * 4);
    int32_t* pos = (int32_t*)p;                p += hgv_align256((size_t)E * 4);
    hipLaunchKernelGGL(k_fill_i32, dim3(grid1d(N)), dim3(256), 0, stream, map, N, -1);
    if (K > 0) hipLaunchKernelGGL(k_mark_perm, dim3(grid1d(K)), dim3(256), 0, stream, perm, K, map);
    hipLaunchKernelGGL(k_edge_flags, dim3(grid1d(E)), dim3(256), 0, stream, edge_index, ld_row, ld_col, E, map, flags);
    int rc = hgv_exclusive_scan_i32(flags, pos, E, p, hgv_exclusive_scan_ws(E), stream);
    if (rc != HGV_OK) return rc;
    hipLaunchKernelGGL(k_edge_compact, dim3(grid1d(E)), dim3(256), 0, stream, edge_index, ld_row, ld_col, E, map, flags,
                       pos, out_ei, ld_out, kept_ids, count, fill_tail);
    HGV_CHECK_LAUNCH("filter_adj");
    return HGV_OK;
}

extern "C" int hgv_gather_rows(const float* x, int64_t ldx, const int64_t* idx, const float* scale, int64_t K, int F,
                               float* out, int64_t ldo, hipStream_t stream) {
    HGV_REQUIRE(K >= 0 && F >= 0, HGV_EINVAL, "gather_rows: negative size");
    if (K == 0 || F == 0) return HGV_OK;
    HGV_REQUIRE(x && idx && out && ldx >= F && ldo >= F, HGV_EINVAL, "gather_rows: null/short buffer");
    const int nchunks = (F + 3) / 4;
    if (hgv_aligned(x, 16) && hgv_aligned(out, 16) && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= 4 * nchunks && ldo >= 4 * nchunks) {
        int ls = 0;
        while ((1 << ls) < nchunks && ls < 4) ++ls;        // up to 16 lanes per row: 35 chunks = 16 lanes x 3 turns
        hipLaunchKernelGGL(k_gather_rows4, dim3((unsigned)hgv_cdiv(K << ls, 256)), dim3(256), 0, stream, x, ldx, idx, scale, K, F,
                           ls, out, ldo);
    } else {
        hipLaunchKernelGGL(k_gather_rows, dim3(grid1d(K * F)), dim3(256), 0, stream, x, ldx, idx, scale, K, F, out, ldo);
    }
    HGV_CHECK_LAUNCH("gather_rows");
    return HGV_OK;
}

extern "C" int hgv_gather_rows_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const int64_t* idx,
                                   const float* scale, int64_t K, int F, float* gx, int64_t ldgx, float* gscale,
                                   hipStream_t stream) {
    HGV_REQUIRE(K >= 0 && F >= 0, HGV_EINVAL, "gather_rows_bwd: negative size");
    if (K == 0 || F == 0) return HGV_OK;
    HGV_REQUIRE(g && idx && ldg >= F && (!gx || ldgx >= F) && (!gscale || (x && ldx >= F)), HGV_EINVAL, "gather_rows_bwd: null/short buffer");
    hipLaunchKernelGGL(k_gather_rows_bwd, dim3((unsigned)hgv_cdiv(K * 64, 256)), dim3(256), 0, stream, g, ldg, x, ldx, idx,
                       scale, K, F, gx, ldgx, gscale);
    HGV_CHECK_LAUNCH("gather_rows_bwd");
    return HGV_OK;
}

extern "C" int hgv_gather_i64(const int64_t* src, const int64_t* idx, int64_t n, int64_t* out, hipStream_t stream) {
    HGV_REQUIRE(n >= 0, HGV_EINVAL, "gather_i64: negative size");
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(src && idx && out, HGV_EINVAL, "gather_i64: null buffer");
    hipLaunchKernelGGL(k_gather_i64, dim3(grid1d(n)), dim3(256), 0, stream, src, idx, n, out);
    HGV_CHECK_LAUNCH("gather_i64");
    return HGV_OK;
}
