// Dense normalised Laplacian of a graph, the matrix behind the reference's spectral terms
// (autoencoder/vae_laplacian.py:24-46 normalized_laplacian, used by l_frequency :170-191, and
// autoencoder/vae_custom_pool.py:19-53 compute_fiedler_value):
//   L = I - diag(dis) A diag(dis),  A[r][c] = number of edges r -> c,  dis = deg^-1/2 over SOURCES (0 for isolated nodes).
// The reference builds A with to_dense_adj and multiplies twice by a diagonal matrix (2 n^3 flops for n^2 useful
// products); here the output buffer first serves as an integer count matrix (cleared, one integer atomic per edge:
// order-independent), then every element is finished in place as delta - (dis[r] * count) * dis[c], the rounding
// order of the two products with a diagonal matrix, so the result equals the reference expression bit for bit.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_lap_count(const int64_t* __restrict__ ei, int64_t ld_row, int64_t ld_col, int64_t E,
                                                   int64_t n, int32_t* __restrict__ cnt, int64_t ld, int32_t* __restrict__ status) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = ei[e * ld_col], c = ei[ld_row + e * ld_col];
        if (r < 0 || r >= n || c < 0 || c >= n) {
            if (status) atomicOr(status, 1);
            continue;
        }
        atomicAdd(cnt + r * ld + c, 1);
    }
}

__global__ __launch_bounds__(256) void k_lap_finish(float* __restrict__ out, int64_t n, int64_t ld, const float* __restrict__ dis) {
    const int64_t r = blockIdx.y;
    const float dr = dis[r];
    int32_t* row_i = reinterpret_cast<int32_t*>(out + r * ld);
    float* row_f = out + r * ld;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ld; c += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (c < n) {
            const float a = (float)row_i[c];
            const float t = __fmul_rn(__fmul_rn(dr, a), dis[c]);
            v = __fsub_rn(c == r ? 1.f : 0.f, t);
        }
        row_f[c] = v;
    }
}

}  // namespace

// edge_index: int64, element (0, e) at ei[e * ld_col], (1, e) at ei[ld_row + e * ld_col]; dis [n] float; out [n, ld] float,
// ld >= n (columns >= n are written as zeros).  status (nullable device int32, caller-zeroed): bit 0 is set when an edge
// names a node outside [0, n) (such edges are skipped; the reference raises ValueError for them).
extern "C" int hgv_normalized_laplacian(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t n,
                                        const float* dis, float* out, int64_t ld, int32_t* status, hipStream_t stream) {
    HGV_REQUIRE(E >= 0 && n >= 0, HGV_EINVAL, "normalized_laplacian: negative size");
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(dis && out && ld >= n && (E == 0 || edge_index), HGV_EINVAL, "normalized_laplacian: null buffer or ld < n");
    HGV_REQUIRE(n <= 65535, HGV_EUNSUPPORTED, "normalized_laplacian: %lld nodes (a dense matrix of more than 65535^2 elements is not built)",
                (long long)n);
    const hipError_t e = hipMemsetAsync(out, 0, (size_t)n * ld * sizeof(float), stream);
    HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "normalized_laplacian: memset failed: %s", hipGetErrorString(e));
    if (E > 0) {
        const int64_t blocks = hgv_cdiv(E, 256);
        hipLaunchKernelGGL(k_lap_count, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, stream, edge_index, ld_row,
                           ld_col, E, n, reinterpret_cast<int32_t*>(out), ld, status);
    }
    const int64_t cb = hgv_cdiv(ld, 256);
    hipLaunchKernelGGL(k_lap_finish, dim3((unsigned)(cb < 64 ? cb : 64), (unsigned)n), dim3(256), 0, stream, out, n, ld, dis);
    HGV_CHECK_LAUNCH("normalized_laplacian");
    return HGV_OK;
}
