// CSR construction from a COO edge_index by a stable LSD radix sort (8 bits per pass).
//
// Contract (bit-exact): perm = argsort_stable(key row), col = other_row[perm],
// rowptr[i] = number of edges whose key < i.  PyG consumes COO directly
// (reference call sites gvae/gvae_skipcon.py:105-118); this structure is what
// lets every aggregation run as an atomics-free segmented reduction.
//
// Pass anatomy (tile = 4096 keys per 256-thread workgroup):
//   hist    : per-tile digit counts in LDS -> counts[digit][tile]
//   scan    : exclusive scan of the digit-major count table (3 small kernels)
//   scatter : each wave owns 1024 consecutive keys, walks them in 16 rounds of 64;
//             in-round rank by 8 ballots (lanes with equal digit), running
//             per-wave digit counters in LDS, then cross-wave prefix -> stable.
// No global atomics anywhere: results are deterministic.
#include "common.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_KPT = 16;
constexpr int RS_TILE = RS_THREADS * RS_KPT;  // 4096
constexpr int RS_WAVE_KEYS = 64 * RS_KPT;     // 1024

__global__ __launch_bounds__(256) void k_extract(const int64_t* __restrict__ ei, int64_t ld_row, int64_t ld_col,
                                                 int64_t E, int64_t N, int by_src, uint32_t* __restrict__ keys,
                                                 int32_t* __restrict__ other, int32_t* __restrict__ status) {
    int bad = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = ei[e * ld_col], d = ei[ld_row + e * ld_col];
        if (s < 0 || s >= N) { bad = 1; s = s < 0 ? 0 : N - 1; }
        if (d < 0 || d >= N) { bad = 1; d = d < 0 ? 0 : N - 1; }
        keys[e] = (uint32_t)(by_src ? s : d);
        other[e] = (int32_t)(by_src ? d : s);
    }
    if (bad && status) status[0] = 1;  // benign race: every writer stores the same value
}

__global__ __launch_bounds__(RS_THREADS) void k_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                     int nb, int32_t* __restrict__ counts) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + w * RS_WAVE_KEYS + lane;
#pragma unroll
    for (int r = 0; r < RS_KPT; ++r) {
        const int64_t i = base + r * 64;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

template <bool IOTA>
__global__ __launch_bounds__(RS_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                        const int32_t* __restrict__ vals_in, int64_t n, int shift,
                                                        int nb, const int32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ keys_out,
                                                        int32_t* __restrict__ vals_out) {
    __shared__ int wcnt[4][256];
    __shared__ int gbase[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) wcnt[i][threadIdx.x] = 0;
    gbase[threadIdx.x] = offsets[(int64_t)threadIdx.x * nb + blockIdx.x];
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * RS_TILE + w * RS_WAVE_KEYS + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t key[RS_KPT];
    int rank[RS_KPT];
#pragma unroll
    for (int r = 0; r < RS_KPT; ++r) {
        const int64_t i = base + r * 64;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0u;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        // m = lanes of this round holding the same digit (valid lanes only)
        int prev = 0;
        if (valid) {
            prev = wcnt[w][d];
        }
        rank[r] = prev + __popcll(m & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lt) == 0ull) wcnt[w][d] = prev + __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // exclusive prefix over the 4 waves, per digit
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = wcnt[i][threadIdx.x];
            wcnt[i][threadIdx.x] = s;
            s += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_KPT; ++r) {
        const int64_t i = base + r * 64;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const int64_t pos = (int64_t)gbase[d] + wcnt[w][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = IOTA ? (int32_t)i : vals_in[i];
        }
    }
}

// ---- exclusive scan of an int32 array (reduce / scan partials / apply) -------------------
constexpr int SC_CHUNK = 256 * 16;

__global__ __launch_bounds__(256) void k_scan_reduce(const int32_t* __restrict__ in, int64_t n,
                                                     int32_t* __restrict__ partial) {
    __shared__ int lds[8];
    const int64_t base = (int64_t)blockIdx.x * SC_CHUNK + (int64_t)threadIdx.x * 16;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    hgv_block_excl_scan_256(s, lds, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_partials(int32_t* __restrict__ partial, int np) {
    __shared__ int lds[8];
    int carry = 0;
    for (int base = 0; base < np; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < np ? partial[i] : 0;
        int tot;
        const int ex = hgv_block_excl_scan_256(v, lds, &tot);
        if (i < np) partial[i] = carry + ex;
        carry += tot;
    }
}

// in and out may be the same array: a thread reads its 16 inputs before it writes them.
__global__ __launch_bounds__(256) void k_scan_apply(const int32_t* in, int64_t n,
                                                    const int32_t* __restrict__ partial, int32_t* out) {
    __shared__ int lds[8];
    const int64_t base = (int64_t)blockIdx.x * SC_CHUNK + (int64_t)threadIdx.x * 16;
    int v[16], s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int run = hgv_block_excl_scan_256(s, lds, &tot) + partial[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

__global__ __launch_bounds__(256) void k_finalize(const uint32_t* __restrict__ keys_sorted,
                                                  const int32_t* __restrict__ vals_sorted,
                                                  const int32_t* __restrict__ other, int64_t E, int64_t N,
                                                  int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                  int32_t* __restrict__ perm) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = t0; e < E; e += stride) {
        const int32_t p = vals_sorted[e];
        perm[e] = p;
        col[e] = other[p];
    }
    for (int64_t i = t0; i <= N; i += stride) {
        // lower bound of key i in the sorted keys
        int64_t lo = 0, hi = E;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys_sorted[mid] < i) lo = mid + 1; else hi = mid;
        }
        rowptr[i] = (int32_t)lo;
    }
}

__global__ void k_zero_i32(int32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}

struct SortWs {
    uint32_t *kA, *kB;
    int32_t *vA, *vB;
    int32_t *counts, *partial;
    size_t bytes;
};

SortWs carve_sort(void* ws, int64_t n) {
    const int64_t nb = hgv_cdiv(n > 0 ? n : 1, RS_TILE);
    const int64_t ncnt = 256 * nb;
    const int64_t np = hgv_cdiv(ncnt, SC_CHUNK);
    char* p = (char*)ws;
    SortWs s;
    size_t off = 0;
    auto take = [&](size_t b) { char* r = p ? p + off : nullptr; off += hgv_align256(b); return r; };
    s.kA = (uint32_t*)take((size_t)n * 4);
    s.kB = (uint32_t*)take((size_t)n * 4);
    s.vA = (int32_t*)take((size_t)n * 4);
    s.vB = (int32_t*)take((size_t)n * 4);
    s.counts = (int32_t*)take((size_t)ncnt * 4);
    s.partial = (int32_t*)take((size_t)np * 4);
    s.bytes = off;
    return s;
}

// Sorts (keys, vals) of length n by key bits [bit_lo, bit_hi), stable.  Input in (k0, v0)
// (v0 == nullptr: vals are the positions 0..n-1).  Ping-pongs between (kA,vA) and (kB,vB);
// returns which buffer holds the result via *res_k / *res_v.
int radix_sort(const uint32_t* k0, const int32_t* v0, int64_t n, int bit_lo, int bit_hi, const SortWs& s,
               hipStream_t st, const uint32_t** res_k, const int32_t** res_v) {
    const int nb = (int)hgv_cdiv(n, RS_TILE);
    const int64_t ncnt = 256 * (int64_t)nb;
    const int np = (int)hgv_cdiv(ncnt, SC_CHUNK);
    const uint32_t* kin = k0;
    const int32_t* vin = v0;
    int flip = (kin == s.kA) ? 1 : 0;
    bool first = true;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        uint32_t* kout = flip ? s.kB : s.kA;
        int32_t* vout = flip ? s.vB : s.vA;
        hipLaunchKernelGGL(k_hist, dim3(nb), dim3(RS_THREADS), 0, st, kin, n, shift, nb, s.counts);
        hipLaunchKernelGGL(k_scan_reduce, dim3(np), dim3(256), 0, st, s.counts, ncnt, s.partial);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, st, s.partial, np);
        hipLaunchKernelGGL(k_scan_apply, dim3(np), dim3(256), 0, st, s.counts, ncnt, s.partial, s.counts);
        if (first && vin == nullptr)
            hipLaunchKernelGGL(k_scatter<true>, dim3(nb), dim3(RS_THREADS), 0, st, kin, vin, n, shift, nb, s.counts, kout, vout);
        else
            hipLaunchKernelGGL(k_scatter<false>, dim3(nb), dim3(RS_THREADS), 0, st, kin, vin, n, shift, nb, s.counts, kout, vout);
        HGV_CHECK_LAUNCH("radix_sort");
        kin = kout; vin = vout; flip ^= 1; first = false;
    }
    *res_k = kin; *res_v = vin;
    return HGV_OK;
}

int bits_for(int64_t n_values) {  // bits needed to represent values in [0, n_values)
    int b = 1;
    while (((int64_t)1 << b) < n_values) ++b;
    return b;
}

}  // namespace

extern "C" size_t hgv_csr_build_ws(int64_t E, int64_t N) {
    (void)N;
    if (E < 0) return 0;
    SortWs s = carve_sort(nullptr, E);
    return s.bytes + hgv_align256((size_t)E * 4) + 256;
}

extern "C" int hgv_csr_build(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t N,
                             int by_src, int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* status, void* ws,
                             size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(N >= 0 && E >= 0, HGV_EINVAL, "csr_build: negative size (E=%lld N=%lld)", (long long)E, (long long)N);
    HGV_REQUIRE(N < ((int64_t)1 << 31) - 1 && E < ((int64_t)1 << 31) - 1, HGV_EUNSUPPORTED,
                "csr_build: E and N must fit int32 (E=%lld N=%lld)", (long long)E, (long long)N);
    HGV_REQUIRE(rowptr != nullptr, HGV_EINVAL, "csr_build: rowptr is null");
    if (E == 0) {
        hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)hgv_cdiv(N + 1, 256)), dim3(256), 0, stream, rowptr, N + 1);
        HGV_CHECK_LAUNCH("csr_build(zero)");
        return HGV_OK;
    }
    HGV_REQUIRE(N > 0, HGV_EINVAL, "csr_build: edges given for an empty node set");
    HGV_REQUIRE(edge_index && col && perm && ws, HGV_EINVAL, "csr_build: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_csr_build_ws(E, N), HGV_EWORKSPACE, "csr_build: workspace %zu < %zu bytes", ws_bytes,
                hgv_csr_build_ws(E, N));
    HGV_REQUIRE(hgv_aligned(ws, 256), HGV_EINVAL, "csr_build: workspace must be 256-byte aligned");
    SortWs s = carve_sort(ws, E);
    int32_t* other = (int32_t*)((char*)ws + s.bytes);
    const int grid = (int)(hgv_cdiv(E, 256) < 4096 ? hgv_cdiv(E, 256) : 4096);
    hipLaunchKernelGGL(k_extract, dim3(grid), dim3(256), 0, stream, edge_index, ld_row, ld_col, E, N, by_src, s.kA,
                       other, status);
    HGV_CHECK_LAUNCH("csr_build(extract)");
    const uint32_t* rk; const int32_t* rv;
    int rc = radix_sort(s.kA, nullptr, E, 0, bits_for(N), s, stream, &rk, &rv);
    if (rc != HGV_OK) return rc;
    const int64_t work = E > N + 1 ? E : N + 1;
    const int fgrid = (int)(hgv_cdiv(work, 256) < 8192 ? hgv_cdiv(work, 256) : 8192);
    hipLaunchKernelGGL(k_finalize, dim3(fgrid), dim3(256), 0, stream, rk, rv, other, E, N, rowptr, col, perm);
    HGV_CHECK_LAUNCH("csr_build(finalize)");
    return HGV_OK;
}

// Stable sort of (uint32 key, int32 value) pairs on key bits [bit_lo, bit_hi); used by the
// per-graph top-k of SAGPooling.  vals_in may be null (values = positions).
extern "C" size_t hgv_sort_pairs_ws(int64_t n) {
    if (n < 0) return 0;
    return carve_sort(nullptr, n).bytes;
}

__global__ void k_copy_pairs(const uint32_t* k, const int32_t* v, int64_t n, uint32_t* ko, int32_t* vo) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (ko) ko[i] = k[i];
        if (vo) vo[i] = v ? v[i] : (int32_t)i;
    }
}

extern "C" int hgv_sort_pairs_u32(const uint32_t* keys_in, const int32_t* vals_in, int64_t n, int bit_lo, int bit_hi,
                                  uint32_t* keys_out, int32_t* vals_out, void* ws, size_t ws_bytes,
                                  hipStream_t stream) {
    HGV_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) - 1, HGV_EINVAL, "sort_pairs: bad n=%lld", (long long)n);
    HGV_REQUIRE(bit_lo >= 0 && bit_hi <= 32 && bit_lo <= bit_hi, HGV_EINVAL, "sort_pairs: bad bit range [%d,%d)", bit_lo, bit_hi);
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(keys_in && ws, HGV_EINVAL, "sort_pairs: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_sort_pairs_ws(n), HGV_EWORKSPACE, "sort_pairs: workspace too small");
    HGV_REQUIRE(hgv_aligned(ws, 256), HGV_EINVAL, "sort_pairs: workspace must be 256-byte aligned");
    SortWs s = carve_sort(ws, n);
    const uint32_t* rk = keys_in; const int32_t* rv = vals_in;
    if (bit_hi > bit_lo) {
        int rc = radix_sort(keys_in, vals_in, n, bit_lo, bit_hi, s, stream, &rk, &rv);
        if (rc != HGV_OK) return rc;
    }
    const int grid = (int)(hgv_cdiv(n, 256) < 4096 ? hgv_cdiv(n, 256) : 4096);
    hipLaunchKernelGGL(k_copy_pairs, dim3(grid), dim3(256), 0, stream, rk, rv, n, keys_out, vals_out);
    HGV_CHECK_LAUNCH("sort_pairs(copy)");
    return HGV_OK;
}

// Exclusive prefix sum of an int32 array (in and out may alias).
extern "C" size_t hgv_exclusive_scan_ws(int64_t n) {
    if (n < 0) return 0;
    return hgv_align256((size_t)hgv_cdiv(n > 0 ? n : 1, SC_CHUNK) * 4);
}

extern "C" int hgv_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
    HGV_REQUIRE(n >= 0, HGV_EINVAL, "exclusive_scan: negative n");
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(in && out && ws, HGV_EINVAL, "exclusive_scan: null buffer");
    HGV_REQUIRE(ws_bytes >= hgv_exclusive_scan_ws(n), HGV_EWORKSPACE, "exclusive_scan: workspace too small");
    const int np = (int)hgv_cdiv(n, SC_CHUNK);
    int32_t* partial = (int32_t*)ws;
    hipLaunchKernelGGL(k_scan_reduce, dim3(np), dim3(256), 0, stream, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, stream, partial, np);
    hipLaunchKernelGGL(k_scan_apply, dim3(np), dim3(256), 0, stream, in, n, partial, out);
    HGV_CHECK_LAUNCH("exclusive_scan");
    return HGV_OK;
}
