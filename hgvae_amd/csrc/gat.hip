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

// The same dot products with 16-byte accesses: a group of L lanes owns a row, lane l keeps the 4-float
// chunks l, l + L, ... (CPL of them) of g[i,:] in registers; neighbour indices are fetched L at a time
// and broadcast; two edges are in flight per group and reduced together.  Needs 16-byte aligned rows.
template <int CPL>
__global__ __launch_bounds__(128) void k_gat_bwd_dot4(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, int64_t N,
                                                      const float* __restrict__ g, int64_t ldg,
                                                      const float* __restrict__ xw, int64_t ldx, int C, int lshift,
                                                      float* __restrict__ dad) {
    const int lane = threadIdx.x & 63;
    const int L = 1 << lshift;
    const int lr = lane & (L - 1);
    const int64_t wave = (int64_t)hgv_xcd_remap(blockIdx.x, gridDim.x) * 2 + (threadIdx.x >> 6);
    const int64_t row = wave * (64 >> lshift) + (lane >> lshift);
    if (row >= N) return;
    const int nchunks = (C + 3) >> 2;
    float4 gr[CPL];
    int coff[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lr + q * L;
        coff[q] = (c < nchunks ? c : 0) * 4;
        float4 v = *reinterpret_cast<const float4*>(g + row * ldg + coff[q]);
        // columns past C (row padding) and chunks past the row count as zeros
        if (c >= nchunks) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (coff[q] + 1 >= C) v.y = 0.f;
        if (coff[q] + 2 >= C) v.z = 0.f;
        if (coff[q] + 3 >= C) v.w = 0.f;
        gr[q] = v;
    }
    const int start = rowptr[row], end = rowptr[row + 1];
    for (int base = start; base < end; base += L) {
        const int k = base + lr;
        const int my_col = k < end ? col[k] : 0;
        const int cnt = min(L, end - base);
        for (int j = 0; j < cnt; j += 2) {
            const float* r0 = xw + (int64_t)__shfl(my_col, j, L) * ldx;
            const float* r1 = xw + (int64_t)__shfl(my_col, min(j + 1, cnt - 1), L) * ldx;
            float4 v0[CPL], v1[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                v0[q] = *reinterpret_cast<const float4*>(r0 + coff[q]);
                v1[q] = *reinterpret_cast<const float4*>(r1 + coff[q]);
            }
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                a0 += gr[q].x * v0[q].x + gr[q].y * v0[q].y + gr[q].z * v0[q].z + gr[q].w * v0[q].w;
                a1 += gr[q].x * v1[q].x + gr[q].y * v1[q].y + gr[q].z * v1[q].z + gr[q].w * v1[q].w;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                if (o < L) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
            }
            if (lr == 0) {
                dad[base + j] = a0;
                if (j + 1 < cnt) dad[base + j + 1] = a1;
            }
        }
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
    const int nchunks = (C + 3) / 4;
    if (hgv_aligned(g, 16) && hgv_aligned(xw, 16) && ldg % 4 == 0 && ldx % 4 == 0 && ldg >= nchunks * 4 && ldx >= nchunks * 4 &&
        nchunks <= 256) {
        // lanes per row (>= 8 unless the row is shorter) x chunks per lane (<= 4): fewest lane slots
        int ls = 0, cpl = 4;
        int64_t best = -1;
        for (int l = 0; l <= 6; ++l) {
            const int Lc = 1 << l;
            if (Lc < 8 && Lc < nchunks) continue;
            int c = (int)hgv_cdiv(nchunks, Lc);
            if (c > 4) continue;
            if (c < 1) c = 1;
            if (best < 0 || (int64_t)Lc * c < best) { best = (int64_t)Lc * c; ls = l; cpl = c; }
        }
        const int rows_per_block = 2 * (64 >> ls);
        const dim3 grid((unsigned)hgv_cdiv(N, rows_per_block)), block(128);
        switch (cpl) {
            case 1: hipLaunchKernelGGL(k_gat_bwd_dot4<1>, grid, block, 0, stream, rowptr, col, N, g, ldg, xw, ldx, C, ls, dad); break;
            case 2: hipLaunchKernelGGL(k_gat_bwd_dot4<2>, grid, block, 0, stream, rowptr, col, N, g, ldg, xw, ldx, C, ls, dad); break;
            case 3: hipLaunchKernelGGL(k_gat_bwd_dot4<3>, grid, block, 0, stream, rowptr, col, N, g, ldg, xw, ldx, C, ls, dad); break;
            default: hipLaunchKernelGGL(k_gat_bwd_dot4<4>, grid, block, 0, stream, rowptr, col, N, g, ldg, xw, ldx, C, ls, dad); break;
        }
    } else {
        int ls = 0;
        while ((1 << ls) < C && ls < 6) ++ls;
        const int rows_per_block = 4 * (64 >> ls);
        hipLaunchKernelGGL(k_gat_bwd_dot, dim3((unsigned)hgv_cdiv(N, rows_per_block)), dim3(256), 0, stream, rowptr, col, N,
                           g, ldg, xw, ldx, C, ls, dad);
    }
    hipLaunchKernelGGL(k_gat_softmax_bwd, dim3((unsigned)hgv_cdiv(N * GL, 256)), dim3(256), 0, stream, rowptr, col, N,
                       a_src, ld_src, a_dst, ld_dst, slope, alpha, alpha_drop, dad, dl, da_dst);
    HGV_CHECK_LAUNCH("gat_bwd");
    return HGV_OK;
}
