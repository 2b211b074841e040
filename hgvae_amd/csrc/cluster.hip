// Pairwise Euclidean distance matrix of the atoms of one structure, the first step of the reference's clustering
// export (gvae/gvae_skipcon.py:352-354: scipy.spatial.distance_matrix(all_coords, all_coords), float64).
// One workgroup per row i: D[i, j] = sqrt(sum_d (c[i,d] - c[j,d])^2) for all j (coalesced float64 stores) and the
// row sum of D, which by symmetry is also the column sum the PCA step centres with - so the matrix is written once
// and never re-read for centring.  No atomics; sums are combined in a fixed order.
#include "common.h"

namespace {

template <int DIM>
__global__ __launch_bounds__(256) void k_distance_rows(const double* __restrict__ c, int64_t n, double* __restrict__ out,
                                                       int64_t ld, double* __restrict__ rowsum) {
    __shared__ double sh[4];
    const int64_t i = blockIdx.x;
    double ci[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) ci[d] = c[i * DIM + d];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double t = ci[d] - c[j * DIM + d];
            s = __dadd_rn(s, __dmul_rn(t, t));
        }
        const double v = sqrt(s);
        out[i * ld + j] = v;
        acc += v;
    }
    acc = hgv_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && rowsum) rowsum[i] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

}  // namespace

// coords [n, dim] float64 (dim 1..4, contiguous) -> out [n, n] float64 (leading dimension ld >= n), rowsum [n] (nullable).
extern "C" int hgv_distance_matrix(const double* coords, int64_t n, int dim, double* out, int64_t ld, double* rowsum,
                                   hipStream_t stream) {
    HGV_REQUIRE(n >= 0 && dim >= 1 && dim <= 4, HGV_EINVAL, "distance_matrix: n >= 0 and 1 <= dim <= 4 expected (n=%lld dim=%d)", (long long)n, dim);
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(coords && out && ld >= n, HGV_EINVAL, "distance_matrix: null buffer or ld < n");
    HGV_REQUIRE(n < ((int64_t)1 << 31), HGV_EUNSUPPORTED, "distance_matrix: too many points");
    const dim3 grid((unsigned)n), block(256);
    switch (dim) {
        case 1: hipLaunchKernelGGL(k_distance_rows<1>, grid, block, 0, stream, coords, n, out, ld, rowsum); break;
        case 2: hipLaunchKernelGGL(k_distance_rows<2>, grid, block, 0, stream, coords, n, out, ld, rowsum); break;
        case 3: hipLaunchKernelGGL(k_distance_rows<3>, grid, block, 0, stream, coords, n, out, ld, rowsum); break;
        default: hipLaunchKernelGGL(k_distance_rows<4>, grid, block, 0, stream, coords, n, out, ld, rowsum); break;
    }
    HGV_CHECK_LAUNCH("distance_matrix");
    return HGV_OK;
}
