// Edge softmax for single-head graph attention, forward and backward, over the by-target CSR
// of the loop-augmented graph (edges without self loops, then one loop per node).
//
// Reference: GATConv(hidden, hidden, heads=1, dropout=0.49) at
// gvae/gvae_skipcon_attention.py:68,109 (PyG defaults negative_slope=0.2, add_self_loops):
//   e_k     = leaky_relu(a_src[col_k] + a_dst[i], slope)          k in row i
//   alpha_k = exp(e_k - max_row e) / (sum_row exp(e - max) + 1e-16)
//   alpha'_k = dropout(alpha_k)    (mask indexed by the position of the edge in the edge LIST)
// The weighted sum out_i = sum_k alpha'_k * xw[col_k] is the aggregation kernel with edge weights.
//
// Rows are handled by groups of 8 lanes (8 rows per wave); a row of any length is strided over
// by its group, reductions by __shfl_xor inside the group.  No atomics.
#include "common.h"

namespace {

constexpr int GL = 8;  // lanes per row

__device__ __forceinline__ float grp_max(float v) {
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, GL));
    return v;
}
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, GL);
    return v;
}
__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

__global__ __launch_bounds__(256) void k_gat_softmax_fwd(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ perm, int64_t N,
                                                         const float* __restrict__ a_src, int64_t lds_,
                                                         const float* __restrict__ a_dst, int64_t ldd_, float slope,
                                                         float p, float scale, const uint8_t* __restrict__ mask,
                                                         int use_philox, uint64_t seed, uint64_t offset0,
                                                         const uint64_t* __restrict__ offset_dev,
                                                         float* __restrict__ alpha, float* __restrict__ alpha_drop) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    const int lr = threadIdx.x & (GL - 1);
    if (row >= N) return;
    const int start = rowptr[row], end = rowptr[row + 1];
    const float ad = a_dst[row * ldd_];
    float m = -INFINITY;
    for (int k = start + lr; k < end; k += GL) m = fmaxf(m, leaky(a_src[(int64_t)col[k] * lds_] + ad, slope));
    m = grp_max(m);
    if (end == start) return;
    float s = 0.f;
    for (int k = start + lr; k < end; k += GL) s += expf(leaky(a_src[(int64_t)col[k] * lds_] + ad, slope) - m);
    s = grp_sum(s) + 1e-16f;
    const HgvPhilox ph(seed);
    const uint64_t offset = offset0 + (offset_dev ? offset_dev[0] : 0ull);
    for (int k = start + lr; k < end; k += GL) {
        const float al = expf(leaky(a_src[(int64_t)col[k] * lds_] + ad, slope) - m) / s;
        alpha[k] = al;
        float keep = 1.f;
        if (p > 0.f) {
            const uint32_t eid = (uint32_t)perm[k];
            if (mask) keep = mask[eid] ? scale : 0.f;
            else if (use_philox) {
                const uint4 r = ph(eid, 0x6A7u, (uint32_t)offset, (uint32_t)(offset >> 32));
                keep = hgv_u01(r.x) >= p ? scale : 0.f;
            }
        }
        alpha_drop[k] = al * keep;
    }
}

// dad[k] = <g[i,:], xw[col_k,:]> for every CSR position k of row i.  A group of L lanes owns a
// row; each lane keeps its 4-float chunks of g[i,:] in registers and the group reduces per edge.
__global__ __launch_bounds__(256) void k_gat_bwd_dot(const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, int64_t N,
                                                     const float* __restrict__ g, int64_t ldg,
                                                     const float* __restrict__ xw, int64_t ldx, int C, int lshift,
                                                     float* __restrict__ dad) {
    const int lane = threadIdx.x & 63;
    const int L = 1 << lshift;
    const int lr = lane & (L - 1);
    const int64_t wave = (int64_t)hgv_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int64_t row = wave * (64 >> lshift) + (lane >> lshift);
    if (row >= N) return;
    const int start = rowptr[row], end = rowptr[row + 1];
    for (int k = start; k < end; ++k) {
        const int64_t j = col[k];
        float acc = 0.f;
        for (int c = lr; c < C; c += L) acc += g[row * ldg + c] * xw[j * ldx + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            if (o < L) acc += __shfl_xor(acc, o, 64);
        if (lr == 0) dad[k] = acc;
    }
}

// softmax + dropout + leaky-relu backward per row:
//   dalpha_k = dad[k] * keep_k ; S = sum_k alpha_k dalpha_k ; de_k = alpha_k (dalpha_k - S)
//   dl_k = de_k * (raw_k > 0 ? 1 : slope) ; da_dst[i] = sum_k dl_k ; dl[k] is scattered to sources later
__global__ __launch_bounds__(256) void k_gat_softmax_bwd(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t N,
                                                         const float* __restrict__ a_src, int64_t lds_,
                                                         const float* __restrict__ a_dst, int64_t ldd_, float slope,
                                                         const float* __restrict__ alpha,
                                                         const float* __restrict__ alpha_drop,
                                                         const float* __restrict__ dad, float* __restrict__ dl,
                                                         float* __restrict__ da_dst) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    const int lr = threadIdx.x & (GL - 1);
    if (row >= N) return;
    const int start = rowptr[row], end = rowptr[row + 1];
    // keep_k = alpha_drop_k / alpha_k is 0 or scale; alpha_k * dalpha_k = alpha_drop_k * dad_k
    float S = 0.f;
    for (int k = start + lr; k < end; k += GL) S += alpha_drop[k] * dad[k];
    S = grp_sum(S);
    const float ad = a_dst[row * ldd_];
    float acc = 0.f;
    for (int k = start + lr; k < end; k += GL) {
        const float raw = a_src[(int64_t)col[k] * lds_] + ad;
        const float de = alpha_drop[k] * dad[k] - alpha[k] * S;
        const float v = de * (raw > 0.f ? 1.f : slope);
        dl[k] = v;
        acc += v;
    }
    acc = grp_sum(acc);
    if (lr == 0) da_dst[row] = acc;
}

}  // namespace

extern "C" int hgv_gat_softmax_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N,
                                   const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst,
                                   float slope, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                                   const uint64_t* offset_dev,
                                   float* alpha, float* alpha_drop, hipStream_t stream) {
    HGV_REQUIRE(N >= 0, HGV_EINVAL, "gat_softmax_fwd: negative N");
    if (N == 0) return HGV_OK;
    HGV_REQUIRE(rowptr && col && perm && a_src && a_dst && alpha && alpha_drop, HGV_EINVAL, "gat_softmax_fwd: null buffer");
    HGV_REQUIRE(p >= 0.f && p < 1.f, HGV_EINVAL, "gat_softmax_fwd: dropout p outside [0,1)");
    const float scale = 1.0f / (float)(1.0 - (double)p);
    const int64_t blocks = hgv_cdiv(N * GL, 256);
    hipLaunchKernelGGL(k_gat_softmax_fwd, dim3((unsigned)blocks), dim3(256), 0, stream, rowptr, col, perm, N, a_src,
                       ld_src, a_dst, ld_dst, slope, p, scale, mask, (mask == nullptr && p > 0.f) ? 1 : 0, seed, offset, offset_dev,
                       alpha, alpha_drop);
    HGV_CHECK_LAUNCH("gat_softmax_fwd");
    return HGV_OK;
}

extern "C" int hgv_gat_bwd(const int32_t* rowptr, const int32_t* col, int64_t N, const float* g, int64_t ldg,
                           const float* xw, int64_t ldx, int C, const float* a_src, int64_t ld_src,
                           const float* a_dst, int64_t ld_dst, float slope, const float* alpha,
                           const float* alpha_drop, float* dad, float* dl, float* da_dst, hipStream_t stream) {
    HGV_REQUIRE(N >= 0 && C > 0, HGV_EINVAL, "gat_bwd: bad sizes");
    if (N == 0) return HGV_OK;
    HGV_REQUIRE(rowptr && col && g && xw && a_src && a_dst && alpha && alpha_drop && dad && dl && da_dst, HGV_EINVAL,
                "gat_bwd: null buffer");
    HGV_REQUIRE(ldg >= C && ldx >= C, HGV_EINVAL, "gat_bwd: leading dimension < C");
    int ls = 0;
    while ((1 << ls) < C && ls < 6) ++ls;
    const int rows_per_block = 4 * (64 >> ls);
    hipLaunchKernelGGL(k_gat_bwd_dot, dim3((unsigned)hgv_cdiv(N, rows_per_block)), dim3(256), 0, stream, rowptr, col, N,
                       g, ldg, xw, ldx, C, ls, dad);
    hipLaunchKernelGGL(k_gat_softmax_bwd, dim3((unsigned)hgv_cdiv(N * GL, 256)), dim3(256), 0, stream, rowptr, col, N,
                       a_src, ld_src, a_dst, ld_dst, slope, alpha, alpha_drop, dad, dl, da_dst);
    HGV_CHECK_LAUNCH("gat_bwd");
    return HGV_OK;
}
