// Projections on the fp16 matrix cores with scaled two-piece operands ("f16x3").
//
// Same operator contract as hgv_gemm_nt16 (gvae/gvae_skipcon.py:61-100):
//   C[M,N] = A1[M,K1] W1^T (+ A2[M,K2] W2^T) (+ bias)
// Every fp32 operand x is scaled by a power of two s so that the largest entry of its row lands in
// [2^12, 2^13) and split into two fp16 pieces y = x s = h + l + r (h = fp16(y), l = fp16(y - h), both
// round-to-nearest: |r| <= 2^-22 |y|).  The product is formed from the three significant piece products
//   al*wh + ah*wl + ah*wh        (each one v_mfma_f32_16x16x32_f16, fp32 accumulation, small terms first)
// and the accumulator is multiplied by 1 / (s_a s_w) at the end, which is exact.  The dropped terms are
// <= 3 * 2^-22 |a w| per product (the bf16 six-product kernel drops 2^-23): half the matrix-core work and a
// quarter of the splitting instructions of that kernel at an error that stays below 1e-6 of the sum of |terms|.
//
// Scales.  W: one scale per output row n, computed by the packing kernel and stored (as its inverse) behind the
// image.  A: one scale per row and tile, chosen by the wave while it streams the row: the scale of a tile starts from
// the largest entry of the row's first 32-wide k chunk and is lowered - accumulators multiplied by the ratio, a power
// of two - whenever a later chunk would exceed 2^15 with it (fp16 overflows at 2^16).  Entries far below the row's
// largest keep an absolute accuracy of 2^-25 (fp16 subnormal spacing) on the 2^12 scale, i.e. 2^-37 of the row maximum.
// Layout of the image, LDS residency, the two-part launch and its pacing are those of the bf16 kernel.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr size_t LDSH_MAX = 160 * 1024;

struct GeoH {
    int N, K1, K2;
    int nb1, nb2;        // 8-column blocks per source
    int chunks;          // 32-wide k chunks over the concatenation
    int tiles;           // n tiles (16 columns each)
    int parts, tpp;      // parts along N, n tiles per part (last part may hold fewer)
    int rows;            // image rows per part: 16 tpp, or N for a single part stored without the dead rows of its last n tile
    int stream;          // 1: the image does not fit even in two parts: 32-wide k chunks are streamed through LDS
};

__host__ __device__ inline size_t part_bytes_h(int rows, int chunks) { return (size_t)chunks * 2 * rows * 64; }
// table + bias row + inverse weight scales live in LDS next to the image
__host__ __device__ inline size_t side_bytes_h(int chunks, int nt) { return (size_t)chunks * 64 + 2 * (size_t)nt * 16 * 4; }

inline bool geo_h(int N, int K1, int K2, GeoH* g) {
    g->N = N; g->K1 = K1; g->K2 = K2;
    g->nb1 = (K1 + 7) / 8; g->nb2 = (K2 + 7) / 8;
    g->chunks = (g->nb1 + g->nb2 + 3) / 4;
    g->tiles = (N + 15) / 16;
    g->stream = 0;
    // whole; whole without the dead rows of the last n tile (138 rows of 9 x 16: the 138+138 -> 138 image then fits and
    // the rows of A are streamed once instead of once per part); two parts along N
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int parts = attempt == 2 ? 2 : 1;
        const int tpp = (g->tiles + parts - 1) / parts;
        const int rows = attempt == 1 ? N : tpp * 16;
        if (tpp <= 9 && part_bytes_h(rows, g->chunks) + side_bytes_h(g->chunks, 9) + 256 <= LDSH_MAX) {
            g->parts = parts; g->tpp = tpp; g->rows = rows;
            return true;
        }
    }
    static const int widths[] = {2, 4, 6, 9, 12, 16, 18};
    for (int w : widths) {
        if (w >= g->tiles && (size_t)2 * 2 * w * 1024 + side_bytes_h(g->chunks, w) + 256 <= LDSH_MAX) {
            g->parts = 1; g->tpp = w; g->rows = w * 16; g->stream = 1;
            return true;
        }
    }
    return false;
}

// scale 2^(12 - floor(log2 amax)) and its inverse; amax >= 0.  Rows of zeros (or below 2^-115) get 2^127 and an
// inverse of zero, infinities the smallest scale.
__device__ __forceinline__ float scale_for(float amax, float& inv) {
    const unsigned e = __float_as_uint(amax) >> 23;
    const unsigned se = min(266u - e, 254u);
    inv = __uint_as_float((254u - se) << 23);
    return __uint_as_float(se << 23);
}

struct PackHArgs {
    const float* W1; int64_t sn1, sk1; const float* W2; int64_t sn2, sk2;
    GeoH g;
    _Float16* out;
    // optional column affine of the operands the image will meet (x_s = a_s * cs_s + ct_s per column k): the image
    // holds W_s diag(cs_s) and bias_out[n] = bias_in[n] + sum_k W1[n,k] ct1[k] + sum_k W2[n,k] ct2[k]
    const float* cs1; const float* ct1; const float* cs2; const float* ct2;
    const float* bias_in; float* bias_out;
    // ind1: source 1 has K1 - 1 weight columns and the image's column K1 - 1 holds sum_k W1[n,k] ct1[k] instead of that
    // sum going into the bias: the operand's last column is an indicator that switches the shift on per row
    int ind1;
};

constexpr int PACKH_MAX_JOBS = 24;
struct PackHJobs {
    PackHArgs job[PACKH_MAX_JOBS];
    int first_block[PACKH_MAX_JOBS + 1];
    int njobs;
};

// One wave per image row (part, r): largest |w| of the row, then the two scaled pieces of every element.
__global__ __launch_bounds__(256) void k_pack16h(const PackHJobs jobs) {
    int q = 0;
#pragma unroll
    for (int j = 1; j < PACKH_MAX_JOBS; ++j)
        if (j < jobs.njobs && (int)blockIdx.x >= jobs.first_block[j]) q = j;
    const PackHArgs& a = jobs.job[q];
    const GeoH& g = a.g;
    const int slots = g.tpp * 16;                                 // output rows per part; the image holds the first g.rows of them
    const int R = ((int)blockIdx.x - jobs.first_block[q]) * 4 + (int)(threadIdx.x >> 6);
    if (R >= g.parts * slots) return;
    const int lane = threadIdx.x & 63;
    const int part = R / slots, r = R - part * slots;
    const int n = R;
    const int rows = g.rows;
    float amax = 0.f;
    double shift1 = 0.0, shift2 = 0.0;                            // sum_k W_s[n,k] ct_s[k], in float64
    const int kw1 = g.K1 - (a.ind1 ? 1 : 0);                      // weight columns of source 1
    if (n < g.N) {
        for (int k = lane; k < kw1; k += 64) {
            const float w = a.W1[(int64_t)n * a.sn1 + (int64_t)k * a.sk1];
            amax = fmaxf(amax, __builtin_fabsf(a.cs1 ? w * a.cs1[k] : w));
            if (a.ct1) shift1 += (double)w * (double)a.ct1[k];
        }
        for (int k = lane; k < g.K2; k += 64) {
            const float w = a.W2[(int64_t)n * a.sn2 + (int64_t)k * a.sk2];
            amax = fmaxf(amax, __builtin_fabsf(a.cs2 ? w * a.cs2[k] : w));
            if (a.ct2) shift2 += (double)w * (double)a.ct2[k];
        }
    }
    float extra = 0.f;                                            // the indicator's weight
    if (a.bias_out != nullptr) {
        shift1 = hgv_wave_sum_d(shift1);
        shift2 = hgv_wave_sum_d(shift2);
        if (a.ind1) extra = (float)shift1;
        if (lane == 0 && n < g.N) a.bias_out[n] = (float)((a.bias_in ? (double)a.bias_in[n] : 0.0) + shift2 + (a.ind1 ? 0.0 : shift1));
    }
    amax = fmaxf(amax, __builtin_fabsf(extra));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    float inv;
    const float s = scale_for(amax, inv);
    const size_t piece_elems = (size_t)rows * 32;
    _Float16* img = a.out + (size_t)part * g.chunks * 2 * piece_elems;
    for (int e = lane; e < (r < rows ? g.chunks * 32 : 0); e += 64) {
        const int c = e >> 5, kk = e & 31;
        const int blk = 4 * c + (kk >> 3), j = kk & 7;
        float x = 0.f;
        if (n < g.N) {
            if (blk < g.nb1) {
                const int k = 8 * blk + j;
                if (k < kw1) { x = a.W1[(int64_t)n * a.sn1 + (int64_t)k * a.sk1]; if (a.cs1) x *= a.cs1[k]; }
                else if (k < g.K1) x = extra;
            } else if (blk < g.nb1 + g.nb2) {
                const int k = 8 * (blk - g.nb1) + j;
                if (k < g.K2) { x = a.W2[(int64_t)n * a.sn2 + (int64_t)k * a.sk2]; if (a.cs2) x *= a.cs2[k]; }
            }
        }
        const float y = x * s;
        const _Float16 h = (_Float16)y;
        const _Float16 l = (_Float16)(y - (float)h);
        const int piece = (kk >> 3) ^ ((r & 8) ? 3 : 0);
        const size_t base = (size_t)c * 2 * piece_elems + (size_t)r * 32 + piece * 8 + j;
        img[base] = h;
        img[base + piece_elems] = l;
    }
    if (lane == 0) {
        float* invw = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(a.out) + (size_t)g.parts * part_bytes_h(g.rows, g.chunks));
        invw[R] = n < g.N ? inv : 0.f;
    }
}

struct NtHArgs {
    const float* A1; int64_t lda1; const float* A2; int64_t lda2;
    GeoH g;
    const _Float16* Wp; const float* bias;
    float* C; int64_t ldc; int64_t M;
    int64_t ntiles;
    int wgs0;
    int paced;
    int pace_slot;                         // which copy of the progress counters a paced launch uses
    const int* gate; int gate_run_if;      // optional device word: the launch does its work only when (*gate != 0) == gate_run_if
    // activated store (kernels instantiated with ACT): C <- dropout(relu(C)), the head of a level's relu -> Dropout ->
    // BatchNorm1d tail (gvae/gvae_skipcon.py:105-107).  Keep decisions come as one bit per element (bit c & 31 of word
    // c >> 5 of the row's act_ldb words; hgv_dropout_bits writes them).  A dropped element is stored as -0.0f when
    // there is no relu in front and a kept zero as +0.0f, so later passes read relu and dropout off the stored values.
    const uint32_t* act_bits; int64_t act_ldb; int act_relu; float act_scale;
    // ... and its column statistics (kernels instantiated with ACT = 2): every workgroup leaves (mean, M2) of the rows it
    // stored in stat_partial[workgroup][2][stat_hp] - workgroup b of B owns the 16-row tiles b, b + B, ... - for the merge
    // that the statistics kernels of the level tail use.  The sums are plain fp32 sums of d and d^2 over the few hundred rows
    // of a wave, added over the waves and turned into M2 = S2 - S1^2 / n in fp64: behind a dropout of rate p the stored values satisfy mean^2 / variance <=
    // (1 - p) / p (the zeros it inserts are part of the variance), so nothing cancels as long as p is not small -
    // callers use this form from p = 0.2 upwards
    float* stat_partial; int stat_hp;
    // backward of the level the product flows into (k_gemm_nt16r instantiated with ACT = 3): the product g = G W is the
    // gradient with respect to the OUTPUT of a level relu? -> Dropout -> BatchNorm1d whose stored input is bwd_d (see act_bits
    // for how relu / dropout are read off it), and what is stored is the gradient with respect to the level's pre-activation,
    //   dy = k * gamma rstd (g - sum(g) / M - xhat sum(g xhat) / M),   xhat = (d - mean) rstd,
    // with the two column sums in bwd_sums[2][stat_hp] (they are known before the product runs: hgv_bn_bwd_sums_alg).
    const float* bwd_d; int64_t bwd_ldd; const float* bwd_gamma; const float* bwd_mean; const float* bwd_rstd; const float* bwd_sums;
    // ACT = 4: the same with g = G W + bwd_add (the level's other consumer has already written its share of the gradient)
    const float* bwd_add; int64_t bwd_ldadd;
    // ACT = 5: the activated store with statistics for TWO dropout draws of the one product (the first level of the two encode
    // passes of a train step, gvae/gvae_skipcon.py:166-171): C and stat_partial take the draw of act_bits, C2 and stat_partial2
    // the draw of act_bits2 (same leading dimensions)
    float* C2; const uint32_t* act_bits2; float* stat_partial2;
};

template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// One step of the reduce-scatter over the 16 lanes that hold the rows of a tile: a lane keeps the half of its values
// that its bit selects (second half when set) and adds the partner's values of the same half; the partner (CTRL) has the
// opposite bit.  x1 / x2: sums and sums of squares, NIN values in, first half NH = ceil(NIN / 2) (the second half is
// padded with zeros).  FIRST: x2 is formed here as the square of x1.
template <int CTRL, int NIN, bool FIRST>
__device__ __forceinline__ void rs_step(float (&x1)[36], float (&x2)[36], bool bit) {
    constexpr int NH = (NIN + 1) / 2;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const float a1 = x1[j], b1 = (NH + j < NIN) ? x1[NH + j] : 0.f;
        const float a2 = FIRST ? a1 * a1 : x2[j], b2 = FIRST ? b1 * b1 : ((NH + j < NIN) ? x2[NH + j] : 0.f);
        const float keep1 = bit ? b1 : a1, send1 = bit ? a1 : b1;
        const float keep2 = bit ? b2 : a2, send2 = bit ? a2 : b2;
        x1[j] = keep1 + dpp_mov<CTRL>(send1);
        x2[j] = keep2 + dpp_mov<CTRL>(send2);
    }
}

// every thread of the launch reads the same word, written by an earlier kernel of the stream: a uniform early exit
#define HGV_GATE_EXIT(a) if ((a).gate != nullptr && ((*(a).gate != 0) ? 1 : 0) != (a).gate_run_if) return

// Progress counters of paced two-part launches (hints only: a stale value costs time, never correctness).
constexpr int PACE_WGS = 256, PACE_WAVES = 16, PACE_SPINS = 512, PACE_AHEAD = 2;
constexpr int64_t PACE_MIN_BYTES = (int64_t)256 << 20;
__device__ int pace_ctr_h[HGV_PACE_SLOTS * 2 * PACE_WGS * PACE_WAVES];

// Waves per workgroup of the 7-9-tile kernels.  Measured on whole 256-graph train steps (same box, ms per step):
// 6: 30.29, 8: 29.90, 10: 30.50, 12: 30.48, 16: 33.2; the depth of the load ring (3-6) changes nothing at any of them.
#ifndef NTH_WAVES_WIDE
#define NTH_WAVES_WIDE 8
#endif
#ifndef NTH_RING_WIDE
#define NTH_RING_WIDE 4
#endif
template <int NT> constexpr int nth_waves() { return NT > 6 ? NTH_WAVES_WIDE : 16; }
template <int NT> constexpr int nth_ring() { return NT > 6 ? NTH_RING_WIDE : 4; }

struct SplitH { f16x8 h, l; };

// largest |x| of the 8 values of this lane and of the three other lanes that hold the same row (lanes i, i+16, i+32, i+48)
__device__ __forceinline__ float row_amax(const f32x4 (&av)[2]) {
    float m = fmaxf(fmaxf(__builtin_fabsf(av[0][0]), __builtin_fabsf(av[0][1])), fmaxf(__builtin_fabsf(av[0][2]), __builtin_fabsf(av[0][3])));
    m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(av[1][0]), __builtin_fabsf(av[1][1])), fmaxf(__builtin_fabsf(av[1][2]), __builtin_fabsf(av[1][3]))));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

__device__ __forceinline__ void split2(float x, float s, _Float16& h, _Float16& l) {
    const float y = x * s;
    h = (_Float16)y;
    l = (_Float16)(y - (float)h);
}

__device__ __forceinline__ SplitH split8h(const f32x4 (&av)[2], float s) {
    SplitH o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 h, l;
        split2(av[j >> 2][j & 3], s, h, l);
        o.h[j] = h; o.l[j] = l;
    }
    return o;
}

template <int NT, int RING, int ACT = 0>
__device__ __forceinline__ void nth_body(const NtHArgs& a, unsigned char* lds_raw, const int part, const int wg_rank, const int wgs) {
    const GeoH& g = a.g;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6), wpb = blockDim.x >> 6;
    const int t0 = part * g.tpp;
    const int rows = g.rows;                                       // rows per piece image
    const size_t img = part_bytes_h(rows, g.chunks);
    int4* tab = reinterpret_cast<int4*>(lds_raw + img);
    float* bias_s = reinterpret_cast<float*>(lds_raw + img + (size_t)g.chunks * 64);
    float* invw_s = bias_s + NT * 16;
    float st1[3] = {0.f, 0.f, 0.f}, st2[3] = {0.f, 0.f, 0.f};      // ACT = 2: this lane's share of the wave's column sums
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)part * img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds_raw);
        const int npieces = (int)(img / 16);
        // batches of 16 loads in flight per thread: with four waves (a few hundred row tiles in all) the 159 KB image is
        // 39 loads per thread, and every batch pays the L2 latency once
        constexpr int CB = 16;
        for (int e0 = t; e0 < npieces; e0 += CB * blockDim.x) {
            f32x4 v[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) v[u] = src[min(e0 + u * (int)blockDim.x, npieces - 1)];
#pragma unroll
            for (int u = 0; u < CB; ++u)
                if (e0 + u * (int)blockDim.x < npieces) dst[e0 + u * (int)blockDim.x] = v[u];
        }
        for (int e = t; e < g.chunks * 4; e += blockDim.x) {
            int second = 0, col = 0; int64_t ld = a.lda1;
            if (e < g.nb1) { col = 8 * e; }
            else if (e < g.nb1 + g.nb2) { second = 1; col = 8 * (e - g.nb1); ld = a.lda2; }
            // both 16-byte halves stay inside the row's leading dimension (columns past K meet zero weights)
            const int c0 = (int)min((int64_t)col, ld - 4), c1 = (int)min((int64_t)col + 4, ld - 4);
            tab[e] = make_int4(c0 * 4, c1 * 4, second, 0);
        }
        const float* invw = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)g.parts * img);
        for (int e = t; e < NT * 16; e += blockDim.x) {
            const int n = t0 * 16 + e;
            bias_s[e] = (a.bias && n < g.N) ? a.bias[n] : 0.f;
            invw_s[e] = invw[t0 * 16 + e];
        }
    }
    __syncthreads();

    const int64_t gw = (int64_t)wib * wgs + wg_rank, nw = (int64_t)wpb * wgs;
    const int64_t my_tiles = gw < a.ntiles ? (a.ntiles - gw + nw - 1) / nw : 0;
    const int steps = g.chunks;

    f32x4 acc[NT];
    // per-lane LDS byte offset of (row i, swizzled piece kq) inside a piece image; the last n tile of an image stored
    // without its dead rows reads the last stored row instead (those outputs are never stored)
    const int wlane = i * 64 + ((kq ^ ((i & 8) ? 3 : 0)) * 16);
    const int row_last = min(16 * (NT - 1) + i, rows - 1);
    const int wlane_last = row_last * 64 + ((kq ^ ((row_last & 8) ? 3 : 0)) * 16);

    // pacing of two-part launches: see k_gemm_nt16_pair
    int* pace_base = pace_ctr_h + a.pace_slot * (2 * PACE_WGS * PACE_WAVES);
    int* pace_mine = pace_base + ((part * PACE_WGS + wg_rank) * PACE_WAVES + wib);
    const int* pace_twin = pace_base + (((part ^ 1) * PACE_WGS + wg_rank) * PACE_WAVES + wib);
    bool pace = ACT == 0 && a.paced != 0;          // the activated stores are one-part launches: never paced
    int pace_step = 0;
    int pace_seen[RING];
#pragma unroll
    for (int d = 0; d < RING; ++d) pace_seen[d] = 0;
    auto pace_gate = [&](int d) __attribute__((always_inline)) {
        if (lane == 0) __hip_atomic_store(pace_mine, pace_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(pace_seen[(d + RING - 2) % RING]) < pace_step - (PACE_AHEAD + 2)) {
            for (int spin = 0;; ++spin) {
                const int seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= pace_step - PACE_AHEAD) break;
                if (spin >= PACE_SPINS) { pace = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        pace_seen[d] = __hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++pace_step;
    };

    int64_t ltile = gw, lidx = 0; int lq = 0;
    const char *lp1, *lp2;
    auto row_pointers = [&]() __attribute__((always_inline)) {
        const int64_t row = min(ltile * 16 + i, a.M - 1);
        lp1 = reinterpret_cast<const char*>(a.A1 + row * a.lda1);
        lp2 = reinterpret_cast<const char*>((a.A2 ? a.A2 : a.A1) + row * (a.A2 ? a.lda2 : a.lda1));
    };
    f32x4 ring[RING][2];
    int4 e_next = tab[kq];
    auto issue = [&](int d) __attribute__((always_inline)) {
        if (pace) pace_gate(d);
        const int4 e = e_next;
        const char* p = e.z ? lp2 : lp1;
        ring[d][0] = *reinterpret_cast<const f32x4*>(p + e.x);
        ring[d][1] = *reinterpret_cast<const f32x4*>(p + e.y);
        if (++lq == steps) {
            lq = 0;
            if (lidx + 1 < my_tiles) { ++lidx; ltile += nw; row_pointers(); }
        }
        e_next = tab[lq * 4 + kq];
    };
    // n tiles in groups of GS: the three piece products of a group are issued term by term across its tiles, so
    // MFMAs that depend on each other are GS instructions apart; the next step's operand is split (with the scale
    // s_next) in the shadow of this step's MFMAs, SPR elements behind every row of MFMAs.
    constexpr int GS = (NT % 3 == 0 || NT > 4) ? 3 : (NT > 1 ? 2 : 1);
    constexpr int MROWS = 3 * ((NT + GS - 1) / GS);
    constexpr int SPR = (8 + MROWS - 1) / MROWS;
    auto compute = [&](const SplitH& cur, const f32x4 (&av_next)[2], int c, float s_next) __attribute__((always_inline)) {
        const unsigned char* base = lds_raw + (size_t)c * 2 * rows * 64;
        SplitH nxt;
        int el = 0;          // compile-time after unrolling: elements of the next operand split so far
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += GS) {
            f16x8 wh[GS], wl[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (n0 + j < NT) {
                    const unsigned char* w = n0 + j == NT - 1 ? base + wlane_last : base + wlane + (size_t)(n0 + j) * 16 * 64;
                    wh[j] = *reinterpret_cast<const f16x8*>(w);
                    wl[j] = *reinterpret_cast<const f16x8*>(w + (size_t)rows * 64);
                }
            }
#define HGV_TERM(WP, AP)                                                                                           \
            _Pragma("unroll") for (int j = 0; j < GS; ++j)                                                        \
                if (n0 + j < NT) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WP[j], AP, acc[n0 + j], 0, 0, 0); \
            _Pragma("unroll") for (int u = 0; u < SPR; ++u) {                                                     \
                if (el < 8) {                                                                                      \
                    _Float16 h_, l_;                                                                               \
                    float x_ = av_next[el >> 2][el & 3];                                                           \
                    asm volatile("" : "+v"(x_));                                                                   \
                    split2(x_, s_next, h_, l_);                                                                    \
                    nxt.h[el] = h_; nxt.l[el] = l_;                                                                \
                }                                                                                                  \
                ++el;                                                                                              \
            }                                                                                                      \
            __builtin_amdgcn_sched_barrier(0x180);
            HGV_TERM(wl, cur.h) HGV_TERM(wh, cur.l) HGV_TERM(wh, cur.h)
#undef HGV_TERM
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= el) {
                _Float16 h_, l_;
                split2(av_next[j >> 2][j & 3], s_next, h_, l_);
                nxt.h[j] = h_; nxt.l[j] = l_;
            }
        }
        return nxt;
    };

    if (my_tiles > 0) {
        row_pointers();
#pragma unroll
        for (int d = 0; d < RING; ++d) issue(d);
    }
    int64_t ctile = gw;
    int cq = 0;
    const int64_t S = my_tiles * steps;
    SplitH cur;
    float s_cur = 1.f, inv_cur = 1.f;          // scale of the lane's row in the tile being accumulated
    if (my_tiles > 0) {
        const f32x4 av0[2] = {ring[0][0], ring[0][1]};
        s_cur = scale_for(row_amax(av0), inv_cur);
        cur = split8h(av0, s_cur);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t u0 = 0; u0 < S; u0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            issue(d);
            if (u0 + d < S) {
                const f32x4 av_next[2] = {ring[(d + 1) % RING][0], ring[(d + 1) % RING][1]};
                // scale of the next step: a new tile starts from its first chunk; inside a tile the scale only ever
                // goes down, when the chunk would leave the fp16 range with the current one
                const bool fresh = cq + 1 == steps;
                const float amax = row_amax(av_next);
                const bool grow = amax * s_cur > 32768.f;
                float inv_t;
                const float s_t = scale_for(amax, inv_t);
                const bool take = fresh || grow;
                const float s_next = take ? s_t : s_cur, inv_next = take ? inv_t : inv_cur;
                cur = compute(cur, av_next, cq, s_next);
                if (fresh) {
                    // weights as the row operand: lane (i, kq) register r = C[row i][16 n + 4 kq + r]
                    const int64_t row = ctile * 16 + i;
                    float u1[36], u2[36];                     // ACT = 2: the stored values (0 for rows past M)
                    if (ACT == 2) {
#pragma unroll
                        for (int q = 0; q < 36; ++q) u1[q] = 0.f;
                    }
                    if (row < a.M) {
                        uint32_t keep_w[(NT + 1) / 2];        // ACT: the row's keep bits, 32 columns per word
                        if (ACT) {
                            const uint32_t* bp = a.act_bits + row * a.act_ldb;
#pragma unroll
                            for (int q = 0; q < (NT + 1) / 2; ++q) keep_w[q] = bp[q];
                        }
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int col = (t0 + n) * 16 + 4 * kq;
                            if (col < a.ldc) {
                                const f32x4 iw = *reinterpret_cast<const f32x4*>(&invw_s[16 * n + 4 * kq]);
                                const f32x4 bs = *reinterpret_cast<const f32x4*>(&bias_s[16 * n + 4 * kq]);
                                f32x4 o;
#pragma unroll
                                for (int r = 0; r < 4; ++r) o[r] = (acc[n][r] * inv_cur) * iw[r] + bs[r];
                                if (ACT) {
                                    const uint32_t nib = keep_w[n >> 1] >> (16 * (n & 1) + 4 * kq);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        float v = a.act_relu ? fmaxf(o[r], 0.f) : o[r];
                                        v *= a.act_scale;
                                        if (v == 0.f) v = 0.f;                       // a kept zero is +0
                                        o[r] = ((nib >> r) & 1u) ? v : (a.act_relu ? 0.f : -0.f);
                                    }
                                }
                                *reinterpret_cast<f32x4*>(a.C + row * a.ldc + col) = o;
                                if (ACT == 2 && NT == 9) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) u1[4 * n + r] = o[r];
                                }
                            }
                        }
                    }
                    if (ACT == 2 && NT == 9) {
                        // column sums over the tile's 16 rows: four reduce-scatter steps (i ^ 8, mirror inside 8, i ^ 2, i ^ 1)
                        rs_step<0x128, 36, true>(u1, u2, (i & 8) != 0);       // row_ror:8
                        rs_step<0x141, 18, false>(u1, u2, (i & 4) != 0);      // row_half_mirror
                        rs_step<0x4E, 9, false>(u1, u2, (i & 2) != 0);        // quad_perm [2,3,0,1]
                        rs_step<0xB1, 5, false>(u1, u2, (i & 1) != 0);        // quad_perm [1,0,3,2]
#pragma unroll
                        for (int q = 0; q < 3; ++q) { st1[q] += u1[q]; st2[q] += u2[q]; }
                    }
                    cq = 0; ctile += nw;
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    ++cq;
                    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                        const float ratio = s_next * inv_cur;          // 1 for the rows that keep their scale
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[n][r] *= ratio;
                    }
                }
                s_cur = s_next; inv_cur = inv_next;
            }
        }
    }
    if (ACT == 0 && a.paced && lane == 0) __hip_atomic_store(pace_mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ACT == 2 && NT == 9) {
        // the waves of a workgroup hold sums for the same columns lane by lane: add them in wave order through LDS (the
        // weight image is dead by now) and leave one (mean, M2) pair per workgroup.  A workgroup owns the 16-row tiles
        // b, b + B, b + 2 B, ... (b = its rank among B workgroups); only the last tile of the matrix may be short.
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);            // [waves][64][6]
#pragma unroll
        for (int q = 0; q < 3; ++q) { red[(wib * 64 + lane) * 6 + q] = st1[q]; red[(wib * 64 + lane) * 6 + 3 + q] = st2[q]; }
        __syncthreads();
        if (wib == 0) {
            // which of the 36 values (v = 4 n + r: column 16 n + 4 kq + r) the lane's three sums belong to: the halves kept above
            const int b3 = (i >> 3) & 1, b2 = (i >> 2) & 1, b1 = (i >> 1) & 1, b0 = i & 1;
            const int len_c = b1 ? 4 : 5;
            const int base = 18 * b3 + 9 * b2 + 5 * b1 + 3 * b0;
            const int len = b0 ? len_c - 3 : 3;
            const int64_t groups = wg_rank < a.ntiles ? (a.ntiles - 1 - wg_rank) / wgs + 1 : 0;
            const double nrows = (double)(groups * 16 - ((groups > 0 && (a.ntiles - 1) % wgs == wg_rank) ? a.ntiles * 16 - a.M : 0));
            float* dst = a.stat_partial + (int64_t)wg_rank * 2 * a.stat_hp;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int v = base + q;
                const int col = 16 * (v >> 2) + 4 * kq + (v & 3);
                if (q < len && col < a.stat_hp) {
                    double s1 = 0.0, s2 = 0.0;
                    for (int w2 = 0; w2 < wpb; ++w2) { s1 += (double)red[(w2 * 64 + lane) * 6 + q]; s2 += (double)red[(w2 * 64 + lane) * 6 + 3 + q]; }
                    dst[col] = groups > 0 ? (float)(s1 / nrows) : 0.f;
                    dst[a.stat_hp + col] = groups > 0 ? (float)(s2 - s1 * s1 / nrows) : 0.f;
                }
            }
        }
    }
}

template <int NT, int ACT = 0>
__global__ __launch_bounds__(64 * nth_waves<NT>()) void k_gemm_nt16h(const NtHArgs a, const int part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT(a);
    nth_body<NT, nth_ring<NT>(), ACT>(a, lds_raw, part, blockIdx.x, gridDim.x);
}

template <int NT0, int NT1>
__global__ __launch_bounds__(64 * nth_waves<NT0>()) void k_gemm_nt16h_pair(const NtHArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT(a);
    if ((int)blockIdx.x < a.wgs0) nth_body<NT0, nth_ring<NT0>()>(a, lds_raw, 0, blockIdx.x, a.wgs0);
    else nth_body<NT1, nth_ring<NT1>()>(a, lds_raw, 1, (int)blockIdx.x - a.wgs0, (int)gridDim.x - a.wgs0);
}

// ------------------------------------------------------------------------------------------
// Weights whose image does not fit in LDS (hidden width 256): 12 waves own one row tile each and all NT n tiles of
// it; the 32-wide k chunks of the image are copied by all threads into one of two LDS buffers while the waves
// multiply from the other (one barrier per chunk).  The scale of a row is fixed per row block the same way as above.
template <int NT>
__global__ __launch_bounds__(768) void k_gemm_nt16hs(const NtHArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT(a);
    const GeoH& g = a.g;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int ROWS = NT * 16, CHUNK_BYTES = 2 * ROWS * 64, UNITS = CHUNK_BYTES / 16, UPT = (UNITS + 767) / 768;
    int4* tab = reinterpret_cast<int4*>(lds_raw + 2 * CHUNK_BYTES);
    float* bias_s = reinterpret_cast<float*>(lds_raw + 2 * CHUNK_BYTES + (size_t)g.chunks * 64);
    float* invw_s = bias_s + ROWS;
    for (int e = t; e < g.chunks * 4; e += 768) {
        int second = 0, col = 0; int64_t ld = a.lda1;
        if (e < g.nb1) { col = 8 * e; }
        else if (e < g.nb1 + g.nb2) { second = 1; col = 8 * (e - g.nb1); ld = a.lda2; }
        const int c0 = (int)min((int64_t)col, ld - 4), c1 = (int)min((int64_t)col + 4, ld - 4);
        tab[e] = make_int4(c0 * 4, c1 * 4, second, 0);
    }
    {
        const float* invw = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.Wp) + part_bytes_h(g.rows, g.chunks));
        for (int e = t; e < ROWS; e += 768) {
            bias_s[e] = (a.bias && e < g.N) ? a.bias[e] : 0.f;
            invw_s[e] = invw[e];
        }
    }
    const f32x4* img = reinterpret_cast<const f32x4*>(a.Wp);
    f32x4 wst[UPT];
    auto w_load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) wst[u] = img[(size_t)c * UNITS + min(t + 768 * u, UNITS - 1)];
    };
    auto w_store = [&](int b) __attribute__((always_inline)) {
        f32x4* dst = reinterpret_cast<f32x4*>(lds_raw + b * CHUNK_BYTES);
#pragma unroll
        for (int u = 0; u < UPT; ++u)
            if (t + 768 * u < UNITS) dst[t + 768 * u] = wst[u];
    };
    const int64_t nblocks = (a.ntiles + 11) / 12;
    const int chunks = g.chunks;
    const int wlane = i * 64 + ((kq ^ ((i & 8) ? 3 : 0)) * 16);

    w_load(0);
    w_store(0);
    __syncthreads();

    f32x4 a_nx[2];
    auto a_load = [&](int64_t rb, int c) __attribute__((always_inline)) {
        const int64_t row = min((rb * 12 + wib) * 16 + i, a.M - 1);
        const int4 e = tab[c * 4 + kq];
        const char* p = e.z ? reinterpret_cast<const char*>(a.A2 + row * a.lda2) : reinterpret_cast<const char*>(a.A1 + row * a.lda1);
        a_nx[0] = *reinterpret_cast<const f32x4*>(p + e.x);
        a_nx[1] = *reinterpret_cast<const f32x4*>(p + e.y);
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < nblocks) a_load(blockIdx.x, 0);
    for (int64_t rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        float s_cur = 1.f, inv_cur = 1.f;
        for (int c = 0; c < chunks; ++c) {
            const int cn = c + 1 < chunks ? c + 1 : 0;
            w_load(cn);
            {
                const float amax = row_amax(a_nx);
                const bool grow = amax * s_cur > 32768.f;
                if (c == 0) {
                    s_cur = scale_for(amax, inv_cur);
                } else if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                    float inv_t;
                    const float s_t = scale_for(amax, inv_t);
                    const float s_next = grow ? s_t : s_cur;
                    const float ratio = s_next * inv_cur;
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[n][r] *= ratio;
                    inv_cur = grow ? inv_t : inv_cur;
                    s_cur = s_next;
                }
            }
            const SplitH cur = split8h(a_nx, s_cur);
            {
                const int64_t rbn = c + 1 < chunks ? rb : (rb + gridDim.x < nblocks ? rb + gridDim.x : rb);
                a_load(rbn, cn);
            }
            const unsigned char* base = lds_raw + buf * CHUNK_BYTES + wlane;
            constexpr int GS = 3;
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += GS) {
                f16x8 wh[GS], wl[GS];
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    if (n0 + j < NT) {
                        const unsigned char* w = base + (size_t)(n0 + j) * 16 * 64;
                        wh[j] = *reinterpret_cast<const f16x8*>(w);
                        wl[j] = *reinterpret_cast<const f16x8*>(w + (size_t)ROWS * 64);
                    }
                }
#define HGV_S_TERM(WP, AP)                                                                                         \
                _Pragma("unroll") for (int j = 0; j < GS; ++j)                                                    \
                    if (n0 + j < NT) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WP[j], AP, acc[n0 + j], 0, 0, 0);
                HGV_S_TERM(wl, cur.h) HGV_S_TERM(wh, cur.l) HGV_S_TERM(wh, cur.h)
#undef HGV_S_TERM
            }
            w_store(buf ^ 1);
            buf ^= 1;
            __syncthreads();
        }
        const int64_t tile = rb * 12 + wib;
        const int64_t row = tile * 16 + i;
        if (tile < a.ntiles && row < a.M) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n * 16 + 4 * kq;
                if (col < a.ldc) {
                    const f32x4 iw = *reinterpret_cast<const f32x4*>(&invw_s[16 * n + 4 * kq]);
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(&bias_s[16 * n + 4 * kq]);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (acc[n][r] * inv_cur) * iw[r] + bs[r];
                    *reinterpret_cast<f32x4*>(a.C + row * a.ldc + col) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weights in registers, operand rows through LDS (k_gemm_nt16r): the form for long operands of the 129...144-column
// products.  With the weight image resident in LDS the rows of A can only be loaded straight into the operand layout -
// 16 rows x 128 bytes per load, every row of a tile nibbled at over the whole time the tile takes - and the launch runs
// at 3.7 TB/s where a walk over the same matrices in whole rows reaches 5.4 (measured without any products: 500 against
// 390 us at 1.26 M rows).  Here wave w of nine keeps the image rows of ITS 16 output columns in registers (8 VGPRs per
// 32-wide k chunk), and a workgroup moves through blocks of 16 (or 32) operand rows: all threads load the block as one
// contiguous range (16 bytes per lane, consecutive lanes consecutive addresses), find the rows' largest entries
// (ds_max_u32 on the float bits), split the scaled values into their two fp16 pieces ONCE and leave them in LDS in the
// layout the matrix cores read (the weight image's: 64 bytes per row and chunk, pieces XOR-swizzled); every wave then
// multiplies the block's row tile(s) against its columns, writes its 16-column piece of the result into a staged output block, and
// all threads store that block as whole rows (activation and column sums on the way, in the row-contiguous layout).
// One barrier per block: the interval between two barriers holds, for consecutive blocks, loads in flight (block b + D),
// row maxima (b + 2), split + stage (b + 1), products (b) and stores (b - 1).
// The scale of a row is exact here (from the largest entry of the whole concatenated row) instead of the running
// scale of the streaming kernels: same error bound, not the same bits.
constexpr int NTR_LDC = 148;           // floats per staged output row: 16-byte writes of 16 rows hit distinct banks
constexpr int NTR_R = 16;              // rows per block
constexpr int NTR_MOVERS = 256;        // four mover waves behind the (up to nine) multiplier waves: 16 lanes per row of the block
constexpr int NTR_THREADS = 576 + NTR_MOVERS;
constexpr int NTR_EPT = 3;             // float4 per mover thread, source and block: lanes 0 ... 11 of a row take f, f + 12, f + 24
#ifndef NTR_SLOTS
#define NTR_SLOTS 2                    // blocks of loaded rows held in registers: one being staged, the others in flight (2, 3
                                       // and 4 take the same time where they fit the registers: the interval, not the memory latency, sets the pace)
#endif
// Smallest row count, in 16-row blocks per compute unit, that takes this kernel.  0: it is the faster one at every row count
// it was measured at (138 rows: 16.4 against 20.4 us, no weight image to copy into LDS; 4 416 rows: 18.2 against 22.2 us;
// 161 k rows: 69 against 75 us; 1.26 M rows: 380 against 560 us - all 138 + 138 -> 138)
#ifndef NTR_MIN_BLOCKS_PER_WG
#define NTR_MIN_BLOCKS_PER_WG 0
#endif

__host__ __device__ constexpr size_t ntr_astage_bytes(int chunks) { return (size_t)chunks * 2 * NTR_R * 64; }
__host__ __device__ constexpr size_t ntr_lds_bytes(int chunks, int act = 0) {
    return 2 * ntr_astage_bytes(chunks) + (size_t)2 * NTR_R * NTR_LDC * 4 + (size_t)2 * NTR_R * 4 + (size_t)4 * NTR_R * 16 * 4 +
           (act == 3 ? (size_t)4 * NTR_R * NTR_LDC * 4 : act == 4 ? (size_t)8 * NTR_R * NTR_LDC * 4 : 0);
}

template <class F, int... Is>
__device__ __forceinline__ void ntr_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }

// EPT: float4 per mover lane, source and block (3: sources up to 144 columns, 6: up to 288).  MAXT: threads of the largest
// launch (registers: 13 waves leave 128 per lane, 12 leave 168).  Outputs past nine column tiles are split over blockIdx.y:
// workgroup (x, y) owns the tiles y T ... y T + T - 1 (T = multiplier waves of the launch) and stages the rows again.
template <int CHUNKS, int NSRC, int ACT, int SLOTS, int EPT = NTR_EPT, int MAXT = NTR_THREADS>
__global__ __launch_bounds__(MAXT) void k_gemm_nt16r(const NtHArgs a) {
    static_assert(SLOTS >= 2 && SLOTS <= 6, "register slots");
    constexpr int R = NTR_R;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT(a);
    const GeoH& g = a.g;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr size_t ASTAGE = ntr_astage_bytes(CHUNKS);
    unsigned char* astage = lds_raw;
    float* cstage = reinterpret_cast<float*>(lds_raw + 2 * ASTAGE);
    float* rowinv = cstage + 2 * R * NTR_LDC;                        // [2][R]: inverse scale of every row of a staged block
    // ACT: keep bits of the rows of the last four blocks, [4][R][8 words].  The movers load them with the rows (their waits on
    // loads are counted ones) and hand them over here: a load inside the multipliers' loop made every interval wait for the
    // completion of the store in front of it
    uint32_t* kbits = reinterpret_cast<uint32_t*>(rowinv + 2 * R);
    constexpr bool BWD = ACT == 3 || ACT == 4, KEEP = ACT == 1 || ACT == 2 || ACT == 5, PAIR = ACT == 5;
    constexpr int KW = 16;                                           // words per row in the ring: 8 per draw
    // ACT = 3: the rows of d that belong to the last four blocks of output, [4][R][NTR_LDC], handed over by the movers likewise
    float* dring = reinterpret_cast<float*>(kbits + 4 * R * KW);
    float* addring = dring + 4 * R * NTR_LDC;                        // ACT = 4: ... and the rows of the addend
    // columns the sources do not reach stay zero in both stage buffers for the whole launch (their weights are zero,
    // but 0 x garbage is not)
    for (int e = t; e < (int)(2 * ASTAGE / 16); e += (int)blockDim.x) reinterpret_cast<f32x4*>(astage)[e] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int64_t nblocks = (a.M + R - 1) / R;
    const int64_t G = gridDim.x;
    const int64_t mine = (int64_t)blockIdx.x < nblocks ? (nblocks - 1 - blockIdx.x) / G + 1 : 0;
    auto block_of = [&](int64_t it) __attribute__((always_inline)) { return (int64_t)blockIdx.x + it * G; };
    const int mwaves = ((int)blockDim.x - NTR_MOVERS) >> 6;          // multiplier waves; the four mover waves follow them
    const int tile0 = (int)blockIdx.y * mwaves;                      // first column tile of this workgroup
    const int ntiles = min(mwaves, g.tiles - tile0);                 // ... and how many it has (waves past them only keep the barriers)
    const int nc4 = (int)(min(a.ldc - 16 * tile0, (int64_t)16 * ntiles) / 4);
    constexpr auto slots = std::make_integer_sequence<int, SLOTS>{};
    // intervals 0 ... mine, rounded up to one + a multiple of SLOTS (the slot of a block is a compile-time index): past the
    // end every step repeats the workgroup's last block - same loads, same staged bytes, same products, same stores
    const int64_t intervals = 1 + (mine + SLOTS - 1) / SLOTS * SLOTS;
    const int64_t last = mine > 0 ? mine - 1 : 0;

    f32x4 st1 = f32x4{0.f, 0.f, 0.f, 0.f}, st2 = f32x4{0.f, 0.f, 0.f, 0.f};        // ACT = 2, 5 (multipliers): sums of what this thread stored
    f32x4 st1b = f32x4{0.f, 0.f, 0.f, 0.f}, st2b = f32x4{0.f, 0.f, 0.f, 0.f};      // ACT = 5: ... of the second draw
    // Every wave passes the same barriers: two in front of the intervals, then one per interval it = 0 ... mine.  Interval it
    // holds, for the multiplier waves, the stores of block it - 1 and the products of block it, next to, for the mover waves,
    // the loads of block it + SLOTS and the scale + split + stage of block it + 1.
    if (w < mwaves) {
        const bool idle = w >= ntiles;
        // ---- multiplier wave w: output columns 16 w ... 16 w + 15.  Lane (i, kq) holds image row 16 w + i, piece kq of every
        // chunk (a last tile stored without its dead rows repeats the last stored row; those outputs meet an inverse scale of zero)
        f16x8 wh[CHUNKS], wl[CHUNKS];
        {
            // tile T of the image: row 16 T + i of a one-part image, row 16 (T mod tpp) + i of part T / tpp of a two-part one
            const int rows = g.rows, T = min(tile0 + w, g.tiles - 1);
            const int part = g.parts == 2 ? T / g.tpp : 0;
            const int r = min(16 * (T - part * g.tpp) + i, rows - 1);
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)part * part_bytes_h(rows, g.chunks) +
                                       (size_t)r * 64 + ((kq ^ ((r & 8) ? 3 : 0)) * 16);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) {
                wh[c] = *reinterpret_cast<const f16x8*>(src + (size_t)c * 2 * rows * 64);
                wl[c] = *reinterpret_cast<const f16x8*>(src + (size_t)c * 2 * rows * 64 + (size_t)rows * 64);
            }
        }
        f32x4 invw4, bias4;
        {
            const float* invw = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)g.parts * part_bytes_h(g.rows, g.chunks));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * min(tile0 + w, g.tiles - 1) + 4 * kq + r;
                invw4[r] = invw[n];
                bias4[r] = (a.bias && n < g.N) ? a.bias[n] : 0.f;
            }
        }
        // store element of this thread: float4 e = t of the block's R x nc4 float4 of output (64 per multiplier wave; threads
        // past the end repeat the last one), rows past the end of the matrix repeat its last row - same values, same address
        const int se = min(t, R * nc4 - 1), srow = se / nc4, scol = 4 * (se % nc4);
        const int soff = srow * (int)a.ldc + 16 * tile0 + scol;
        auto rows_over = [&](int64_t r0) __attribute__((always_inline)) {
            const int d = (int)max(r0 - (a.M - 1), (int64_t)-R);        // uniform: - R ... 0
            return max(d + srow, 0);
        };
        const int dropped = a.act_relu ? 0 : (int)0x80000000u;
        // ACT = 3: the constants of this thread's four columns (columns past N store zeros)
        f32x4 bw_a = f32x4{0.f, 0.f, 0.f, 0.f}, bw_mu = bw_a, bw_rs = bw_a, bw_sg = bw_a, bw_sgx = bw_a;
        int bw_valid = 0;
        if (BWD) {
            const float invn = 1.f / (float)a.M;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * tile0 + scol + r;
                if (c < g.N) {
                    const float rs = a.bwd_rstd[c];
                    bw_valid |= 1 << r;
                    bw_a[r] = (a.bwd_gamma ? a.bwd_gamma[c] : 1.f) * rs;
                    bw_mu[r] = a.bwd_mean[c];
                    bw_rs[r] = rs;
                    bw_sg[r] = a.bwd_sums[c] * invn;
                    bw_sgx[r] = a.bwd_sums[a.stat_hp + c] * invn;
                }
            }
        }
        __syncthreads();
        __syncthreads();
        const int qoff = i * 64 + ((kq ^ ((i & 8) ? 3 : 0)) * 16);
        for (int64_t it = 0; it < intervals; ++it) {
            if (it >= 1) {          // block it - 1 (the workgroup's last one again past the end): staged products -> rows of C
                const int64_t r0 = block_of(min(it - 1, last)) * R;
                const int over = rows_over(r0);
                f32x4 o = *reinterpret_cast<const f32x4*>(cstage + ((it - 1) & 1) * R * NTR_LDC + srow * NTR_LDC + scol);
                if (BWD) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dring + ((it - 1) & 3) * R * NTR_LDC + srow * NTR_LDC + scol);
                    if (ACT == 4) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(addring + ((it - 1) & 3) * R * NTR_LDC + srow * NTR_LDC + scol);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += av[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (dv[r] - bw_mu[r]) * bw_rs[r];
                        float dd = bw_a[r] * (o[r] - bw_sg[r] - xh * bw_sgx[r]);
                        dd *= a.act_scale;
                        const bool dead = a.act_relu ? !(dv[r] > 0.f) : __float_as_uint(dv[r]) == 0x80000000u;
                        o[r] = (dead || !(bw_valid & (1 << r))) ? 0.f : dd;
                    }
                } else if (KEEP) {
                    f32x4 o2 = o;
                    if (PAIR) {          // the second draw of the same values
                        const uint32_t nib2 = kbits[(((it - 1) & 3) * R + srow) * KW + 8 + (scol >> 5)] >> (scol & 31);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int keep = __builtin_amdgcn_sbfe(nib2, r, 1);
                            o2[r] = __int_as_float((__float_as_int(o2[r]) & keep) | (dropped & ~keep));
                        }
                    }
                    const uint32_t nib = kbits[(((it - 1) & 3) * R + srow) * KW + (scol >> 5)] >> (scol & 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {          // keep bit -> all-ones / zero mask (v_bfe_i32), then a bit select (v_bfi_b32)
                        const int keep = __builtin_amdgcn_sbfe(nib, r, 1);
                        o[r] = __int_as_float((__float_as_int(o[r]) & keep) | (dropped & ~keep));
                    }
                    if (PAIR) {
                        *reinterpret_cast<f32x4*>(a.C2 + r0 * a.ldc + (unsigned)(soff - over * (int)a.ldc)) = o2;
                        const bool counts = it - 1 <= last && over == 0 && t < R * nc4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = counts ? o2[r] : 0.f; st1b[r] += d; st2b[r] += d * d; }
                    }
                }
                *reinterpret_cast<f32x4*>(a.C + r0 * a.ldc + (unsigned)(soff - over * (int)a.ldc)) = o;
                if (ACT == 2 || PAIR) {
                    const bool counts = it - 1 <= last && over == 0 && t < R * nc4;          // the first and only time this element is stored
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = counts ? o[r] : 0.f; st1[r] += d; st2[r] += d * d; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // the store leaves before the products start
            if (!idle) {
                const unsigned char* q = astage + (it & 1) * ASTAGE + qoff;
                float* dst = cstage + (it & 1) * R * NTR_LDC;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                f16x8 ah[2], al[2];
                ah[0] = *reinterpret_cast<const f16x8*>(q);
                al[0] = *reinterpret_cast<const f16x8*>(q + R * 64);
                const float inv = rowinv[(it & 1) * R + i];
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c) {
                    if (c + 1 < CHUNKS) {
                        ah[(c + 1) & 1] = *reinterpret_cast<const f16x8*>(q + (c + 1) * 2 * R * 64);
                        al[(c + 1) & 1] = *reinterpret_cast<const f16x8*>(q + (c + 1) * 2 * R * 64 + R * 64);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], ah[c & 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], al[c & 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], ah[c & 1], acc, 0, 0, 0);
                }
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc[r] * inv) * invw4[r] + bias4[r];
                if (KEEP) {          // relu and the dropout scale here, the keep decision where the row is stored
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = a.act_relu ? fmaxf(o[r], 0.f) : o[r];
                        v *= a.act_scale;
                        o[r] = v == 0.f ? 0.f : v;                   // a kept zero is +0
                    }
                }
                *reinterpret_cast<f32x4*>(dst + i * NTR_LDC + 16 * w + 4 * kq) = o;
            }
            __syncthreads();
        }
    } else {
        // ---- mover thread (row mr of the block, lane mj of the row's 16): float4 f = min(mj, 11) + 12 u (u = 0, 1, 2) of the row,
        // the last one repeated where the row is shorter - same loads, same staged bytes; lanes 12 ... 15 repeat lane 11
        const int tl = t - mwaves * 64, mr = tl >> 4, mj = min(tl & 15, 11);
        int eoff[NSRC][EPT], ewr[NSRC][EPT];          // element offset (floats) from the block's first row; byte offset of its h halves in a stage buffer
#pragma unroll
        for (int sidx = 0; sidx < NSRC; ++sidx) {
            const int nf4 = ((sidx ? g.K2 : g.K1) + 3) / 4;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int f = min(mj + 12 * u, nf4 - 1);
                eoff[sidx][u] = mr * (int)(sidx ? a.lda2 : a.lda1) + 4 * f;
                const int kc = (sidx ? 8 * g.nb1 : 0) + 4 * f;            // column of the concatenation
                const int c = kc >> 5, c4 = (kc >> 2) & 7;          // piece c4 >> 1 of the chunk (its columns 8 q ... 8 q + 7), half c4 & 1
                ewr[sidx][u] = ((c * 2) * R + mr) * 64 + (((c4 >> 1) ^ ((mr & 8) ? 3 : 0)) * 16) + (c4 & 1) * 8;
            }
        }
        f32x4 raw[SLOTS][NSRC][EPT];
        f32x4 draw[SLOTS][EPT];                       // ACT = 3, 4: this lane's part of the row of d
        f32x4 aadd[SLOTS][EPT];                       // ACT = 4: ... and of the addend
        int eoffd[EPT], eoffa[EPT], dwr[EPT];
        if (BWD) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int f = min(mj + 12 * u, nc4 - 1);
                eoffd[u] = mr * (int)a.bwd_ldd + 4 * f;
                if (ACT == 4) eoffa[u] = mr * (int)a.bwd_ldadd + 4 * f;
                dwr[u] = mr * NTR_LDC + 4 * f;
            }
        }
        uint32_t kb[SLOTS];                           // ACT: word min(lane of the row, 7) of the row's keep bits
        // lane j of the row's sixteen: word min(j, 7) of the draw's keep bits (ACT = 5: lanes 8 ... 15 word j - 8 of the second draw's)
        const int kword = KEEP ? min(PAIR ? (tl & 7) : min(tl & 15, 7), (int)a.act_ldb - 1) : 0;
        const uint32_t* kbase = (PAIR && (tl & 8)) ? a.act_bits2 : a.act_bits;
        // rows past the end of the matrix (last block of the launch) repeat its last row.  rows_over: rows of this thread past
        // the last row when its block starts at r0 (scalar part first: two VALU instructions)
        auto rows_over = [&](int64_t r0) __attribute__((always_inline)) {
            const int d = (int)max(r0 - (a.M - 1), (int64_t)-R);        // uniform: - R ... 0
            return max(d + mr, 0);
        };
        auto gload = [&](auto slot, int64_t it) __attribute__((always_inline)) {
            constexpr int S = decltype(slot)::value;
            const int64_t r0 = block_of(min(it, last)) * R;
            const int over = rows_over(r0);
#pragma unroll
            for (int sidx = 0; sidx < NSRC; ++sidx) {
                const float* base = sidx ? a.A2 + r0 * a.lda2 : a.A1 + r0 * a.lda1;          // uniform base + 32-bit lane offset
                const int back = over * (int)(sidx ? a.lda2 : a.lda1);
#pragma unroll
                for (int u = 0; u < EPT; ++u) raw[S][sidx][u] = *reinterpret_cast<const f32x4*>(base + (unsigned)(eoff[sidx][u] - back));
            }
            if (BWD) {
                const float* base = a.bwd_d + r0 * a.bwd_ldd;
                const int back = over * (int)a.bwd_ldd;
#pragma unroll
                for (int u = 0; u < EPT; ++u) draw[S][u] = *reinterpret_cast<const f32x4*>(base + (unsigned)(eoffd[u] - back));
                if (ACT == 4) {
                    const float* abase = a.bwd_add + r0 * a.bwd_ldadd;
                    const int aback = over * (int)a.bwd_ldadd;
#pragma unroll
                    for (int u = 0; u < EPT; ++u) aadd[S][u] = *reinterpret_cast<const f32x4*>(abase + (unsigned)(eoffa[u] - aback));
                }
            } else if (KEEP) kb[S] = (kbase + r0 * a.act_ldb)[(unsigned)((mr - over) * (int)a.act_ldb + kword)];
        };
        // scale of the row from its largest entry (the row's 16 lanes hold all of it), split, stage; the inverse scale for the multipliers
        auto stage = [&](auto slot, int64_t it) __attribute__((always_inline)) {
            constexpr int S = decltype(slot)::value;
            unsigned char* dst = astage + (it & 1) * ASTAGE;
            // the slot's values exist from here on: without this the compiler starts on the row maxima one interval early, right
            // behind the loads, and waits for them there
#pragma unroll
            for (int sidx = 0; sidx < NSRC; ++sidx)
#pragma unroll
                for (int u = 0; u < EPT; ++u) asm volatile("" : "+v"(raw[S][sidx][u]));
            if (BWD) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) asm volatile("" : "+v"(draw[S][u]));
                if (ACT == 4) {
#pragma unroll
                    for (int u = 0; u < EPT; ++u) asm volatile("" : "+v"(aadd[S][u]));
                }
            } else if (KEEP) asm volatile("" : "+v"(kb[S]));
            float m = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < NSRC; ++sidx)
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const f32x4 v = raw[S][sidx][u];
                    m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
                }
            m = fmaxf(m, dpp_mov<0x128>(m));          // row_ror:8, 4, 2, 1: every lane of the row ends with the row's maximum
            m = fmaxf(m, dpp_mov<0x124>(m));
            m = fmaxf(m, dpp_mov<0x122>(m));
            m = fmaxf(m, dpp_mov<0x121>(m));
            float inv;
            const float sc = scale_for(m, inv);
            rowinv[(it & 1) * R + mr] = inv;
            if (BWD) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) *reinterpret_cast<f32x4*>(dring + (it & 3) * R * NTR_LDC + dwr[u]) = draw[S][u];
                if (ACT == 4) {
#pragma unroll
                    for (int u = 0; u < EPT; ++u) *reinterpret_cast<f32x4*>(addring + (it & 3) * R * NTR_LDC + dwr[u]) = aadd[S][u];
                }
            } else if (KEEP) kbits[((it & 3) * R + mr) * KW + (PAIR ? (tl & 15) : min(tl & 15, 7))] = kb[S];
            // elements two at a time: the split of eight values compiles to packed multiplies / conversions
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            constexpr int NE = NSRC * EPT;
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += 2) {
                const int e1 = e0 + 1 < NE ? e0 + 1 : e0;
                const f32x4 av[2] = {raw[S][e0 / EPT][e0 % EPT], raw[S][e1 / EPT][e1 % EPT]};
                const SplitH o = split8h(av, sc);
                *reinterpret_cast<f16x4*>(dst + ewr[e0 / EPT][e0 % EPT]) = f16x4{o.h[0], o.h[1], o.h[2], o.h[3]};
                *reinterpret_cast<f16x4*>(dst + ewr[e0 / EPT][e0 % EPT] + R * 64) = f16x4{o.l[0], o.l[1], o.l[2], o.l[3]};
                if (e1 != e0) {
                    *reinterpret_cast<f16x4*>(dst + ewr[e1 / EPT][e1 % EPT]) = f16x4{o.h[4], o.h[5], o.h[6], o.h[7]};
                    *reinterpret_cast<f16x4*>(dst + ewr[e1 / EPT][e1 % EPT] + R * 64) = f16x4{o.l[4], o.l[5], o.l[6], o.l[7]};
                }
                __builtin_amdgcn_sched_barrier(0);      // pair by pair: interleaved, the temporaries of all pairs do not fit next to the slots
            }
        };
        ntr_for([&](auto k) __attribute__((always_inline)) { gload(k, decltype(k)::value); }, slots);
        __syncthreads();
        stage(std::integral_constant<int, 0>{}, 0);
        __syncthreads();
        // The interval has no branches, so the waits on the loads stay counted (behind a branch every wait becomes vmcnt(0):
        // nothing would stay in flight across a barrier).  Slot k = it mod SLOTS held block it, staged one interval ago.
        auto interval = [&](auto k, int64_t it) __attribute__((always_inline)) {
            constexpr int K = decltype(k)::value;
            gload(k, it + SLOTS);
            __builtin_amdgcn_sched_barrier(0);          // the loads leave first
            stage(std::integral_constant<int, (K + 1) % SLOTS>{}, it + 1);
            __builtin_amdgcn_sched_barrier(0);          // nothing of the next interval (its stage step waits for loads) moves up
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        interval(std::integral_constant<int, 0>{}, 0);
        for (int64_t it = 1; it < intervals; it += SLOTS)
            ntr_for([&](auto k) __attribute__((always_inline)) { interval(std::integral_constant<int, (decltype(k)::value + 1) % SLOTS>{}, it + decltype(k)::value); }, slots);
    }
    constexpr int NSETS = PAIR ? 2 : (ACT == 2 ? 1 : 0);
#pragma unroll
    for (int set = 0; set < NSETS; ++set) {
        // the stage buffers are dead: every multiplier thread's sums, then per column the sums of the 16 threads that stored it,
        // added in row order; workgroup b of G holds the 16-row groups b, b + G, ...
        float* red = reinterpret_cast<float*>(astage);
        if (set) __syncthreads();          // the first draw's sums have been read
        if (w < mwaves) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { red[t * 8 + r] = set ? st1b[r] : st1[r]; red[t * 8 + 4 + r] = set ? st2b[r] : st2[r]; }
        }
        __syncthreads();
        const int ncol = 4 * nc4;
        if (t < ncol) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < R; ++r) {
                s1 += (double)red[((t >> 2) + nc4 * r) * 8 + (t & 3)];
                s2 += (double)red[((t >> 2) + nc4 * r) * 8 + 4 + (t & 3)];
            }
            const int64_t part = blockIdx.x;
            const int64_t groups = part < nblocks ? (nblocks - 1 - part) / G + 1 : 0;
            const double nrows = (double)(groups * 16 - ((groups > 0 && (nblocks - 1) % G == part) ? nblocks * 16 - a.M : 0));
            float* dst = (set ? a.stat_partial2 : a.stat_partial) + part * 2 * a.stat_hp;
            if (t < a.stat_hp) {
                dst[t] = groups > 0 ? (float)(s1 / nrows) : 0.f;
                dst[a.stat_hp + t] = groups > 0 ? (float)(s2 - s1 * s1 / nrows) : 0.f;
            }
        }
    }
}

int cu_count_h() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

template <typename Kern>
int raise_lds_h(Kern kern) {
    static bool raised[64] = {false};          // one instance per kernel type
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)LDSH_MAX);
        HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "gemm_nt16h: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        raised[dev] = true;
    }
    return HGV_OK;
}

constexpr int nth_min_waves() { return 4; }

// workgroups and waves per workgroup of a one-part launch over ntiles row tiles
template <int NT>
void nth_grid(int64_t ntiles, int64_t* wpb_out, int64_t* blocks_out) {
    const int cus = cu_count_h();
    int64_t wpb = hgv_cdiv(ntiles, cus);
    if (wpb < nth_min_waves()) wpb = nth_min_waves();
    if (wpb > nth_waves<NT>()) wpb = nth_waves<NT>();
    int64_t blocks = hgv_cdiv(ntiles, wpb);
    if (blocks > cus) blocks = cus;
    *wpb_out = wpb; *blocks_out = blocks;
}

template <int NT, int ACT = 0>
int launch_nth(const NtHArgs& a, int part, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt16h<NT, ACT>;
    const int rc = raise_lds_h(kern);
    if (rc != HGV_OK) return rc;
    int64_t wpb, blocks;
    nth_grid<NT>(a.ntiles, &wpb, &blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((unsigned)(64 * wpb)), lds, st, a, part);
    return HGV_OK;
}

int launch_nth_part(const NtHArgs& a, int part, int nt, size_t lds, hipStream_t st) {
    if (a.act_bits && a.stat_partial) return nt == 9 ? launch_nth<9, 2>(a, part, lds, st) : HGV_EUNSUPPORTED;
    if (a.act_bits) return nt == 9 ? launch_nth<9, 1>(a, part, lds, st) : HGV_EUNSUPPORTED;
    switch (nt) {
        case 1: return launch_nth<1>(a, part, lds, st);
        case 2: return launch_nth<2>(a, part, lds, st);
        case 3: return launch_nth<3>(a, part, lds, st);
        case 4: return launch_nth<4>(a, part, lds, st);
        case 5: return launch_nth<5>(a, part, lds, st);
        case 6: return launch_nth<6>(a, part, lds, st);
        case 7: return launch_nth<7>(a, part, lds, st);
        case 8: return launch_nth<8>(a, part, lds, st);
        default: return launch_nth<9>(a, part, lds, st);
    }
}

#ifndef NTR_ENABLE
#define NTR_ENABLE 1
#endif
// The register-weight kernel covers 3 ... 18 column tiles (past nine they are split over two workgroups per row block),
// four to nine k chunks from sources of at most 144 columns, and the two wide shapes of the hidden-width-256 model (256 -> N
// and 256 + 256 -> N: 8 / 16 chunks, six float4 per mover lane and source, eight multiplier waves at most so that twelve waves
// share the registers); the activated stores are built for the two shapes of the level tails (129...144 -> 129...144 and
// 138 + 138 -> 129...144 columns).
struct NtrPlan { bool ok; bool wide; int vparts, mwaves; };
NtrPlan ntr_plan(const GeoH& g, int64_t M, bool two, bool act, bool pair = false) {
    NtrPlan p{false, false, 1, 0};
    if (!NTR_ENABLE || g.tiles < 3 || g.tiles > 18) return p;
    p.wide = g.K1 > 144 || g.K2 > 144;
    if (p.wide) {
        if (!((two && g.K1 == 256 && g.K2 == 256) || (!two && g.K1 == 256))) return p;
    } else {
        // short sources leave the movers (three float4 per lane and source, whatever its width) and the barrier per block with too
        // little to amortise: 28 -> 138 at 1.26 M rows 286 against 252 us, 28 + 28 -> 138 334 against 277, 96 -> 138 304 against 325
        if (g.chunks < 4 || g.chunks > 9 || (!two && g.chunks > 5)) return p;
    }
    const int per = p.wide ? 8 : 9;
    p.vparts = (g.tiles + per - 1) / per;
    p.mwaves = (g.tiles + p.vparts - 1) / p.vparts;
    if (act && !pair && (p.vparts != 1 || p.wide || g.parts != 1 || !((g.chunks == 5 && !two) || (g.chunks == 9 && two)))) return p;
    if (pair && (p.vparts != 1 || p.wide || g.parts != 1 || g.tiles != 9 || !(g.chunks == 7 && two))) return p;          // 97...112 + 97...112 -> 129...144
    if (hgv_cdiv(M, NTR_R) < (int64_t)NTR_MIN_BLOCKS_PER_WG * cu_count_h()) return p;
    p.ok = true;
    return p;
}
#ifndef NTR_WGS_LIGHT
#define NTR_WGS_LIGHT 1
#endif
// workgroups of a launch: one per compute unit (and column split).  The one-source product without activation needs few enough
// registers (72) and LDS for two of them to share a compute unit: measured 3 % slower than one (320 against 309 us at 1.26 M rows)
int64_t ntr_grid(int64_t M, int vparts = 1, bool light = false) {
    const int64_t nblocks = hgv_cdiv(M, NTR_R);
    int64_t wgs = (int64_t)cu_count_h() * (light ? NTR_WGS_LIGHT : 1) / vparts;
    if (wgs < 1) wgs = 1;
    return nblocks < wgs ? nblocks : wgs;
}
template <int CHUNKS, int NSRC, int ACT, int EPT = NTR_EPT, int MAXT = NTR_THREADS>
int launch_ntr_k(const NtHArgs& a, const NtrPlan& p, hipStream_t st) {
    auto kern = k_gemm_nt16r<CHUNKS, NSRC, ACT, NTR_SLOTS, EPT, MAXT>;
    const int rc = raise_lds_h(kern);
    if (rc != HGV_OK) return rc;
    HGV_REQUIRE(64 * p.mwaves + NTR_MOVERS <= MAXT, HGV_ELAUNCH, "gemm_nt16h: %d multiplier waves do not fit this kernel", p.mwaves);
    hipLaunchKernelGGL(kern, dim3((unsigned)ntr_grid(a.M, p.vparts, CHUNKS <= 5 && ACT == 0), (unsigned)p.vparts),
                       dim3((unsigned)(64 * p.mwaves + NTR_MOVERS)), ntr_lds_bytes(CHUNKS, ACT), st, a);
    return HGV_OK;
}
// shapes whose product can store the backward of the level it flows into
bool ntr_bwd_shape(const GeoH& g, const NtrPlan& p, bool two) {
    return p.ok && !p.wide && p.vparts == 1 && g.N <= 144 && (two ? (g.chunks == 4 || g.chunks == 9) : (g.chunks == 4 || g.chunks == 5));
}
int launch_ntr(const NtHArgs& a, const NtrPlan& p, hipStream_t st) {
    if (a.act_bits2) return launch_ntr_k<7, 2, 5>(a, p, st);
    if (a.bwd_d && a.bwd_add) {
        if (a.A2) return a.g.chunks == 9 ? launch_ntr_k<9, 2, 4>(a, p, st) : HGV_EUNSUPPORTED;
        return a.g.chunks == 5 ? launch_ntr_k<5, 1, 4>(a, p, st) : HGV_EUNSUPPORTED;
    }
    if (a.bwd_d) {
        if (a.A2) return a.g.chunks == 4 ? launch_ntr_k<4, 2, 3>(a, p, st) : launch_ntr_k<9, 2, 3>(a, p, st);
        return a.g.chunks == 4 ? launch_ntr_k<4, 1, 3>(a, p, st) : launch_ntr_k<5, 1, 3>(a, p, st);
    }
    if (a.act_bits) {
        const bool stats = a.stat_partial != nullptr;
        if (a.A2) return stats ? launch_ntr_k<9, 2, 2>(a, p, st) : launch_ntr_k<9, 2, 1>(a, p, st);
        return stats ? launch_ntr_k<5, 1, 2>(a, p, st) : launch_ntr_k<5, 1, 1>(a, p, st);
    }
    if (p.wide) return a.A2 ? launch_ntr_k<16, 2, 0, 6, 768>(a, p, st) : launch_ntr_k<8, 1, 0, 6, 768>(a, p, st);
    if (a.A2) {
        switch (a.g.chunks) {
            case 4: return launch_ntr_k<4, 2, 0>(a, p, st);
            case 5: return launch_ntr_k<5, 2, 0>(a, p, st);
            case 6: return launch_ntr_k<6, 2, 0>(a, p, st);
            case 7: return launch_ntr_k<7, 2, 0>(a, p, st);
            case 8: return launch_ntr_k<8, 2, 0>(a, p, st);
            default: return launch_ntr_k<9, 2, 0>(a, p, st);
        }
    }
    return a.g.chunks == 4 ? launch_ntr_k<4, 1, 0>(a, p, st) : launch_ntr_k<5, 1, 0>(a, p, st);
}

template <int NT>
int launch_nths_nt(const NtHArgs& a, hipStream_t st) {
    auto kern = k_gemm_nt16hs<NT>;
    const int rc = raise_lds_h(kern);
    if (rc != HGV_OK) return rc;
    const size_t lds = (size_t)2 * 2 * NT * 1024 + side_bytes_h(a.g.chunks, NT);
    int64_t blocks = hgv_cdiv(a.ntiles, 12);
    if (blocks > cu_count_h()) blocks = cu_count_h();
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(768), lds, st, a);
    return HGV_OK;
}

int launch_nths(const NtHArgs& a, hipStream_t st) {
    switch (a.g.tpp) {
        case 2: return launch_nths_nt<2>(a, st);
        case 4: return launch_nths_nt<4>(a, st);
        case 6: return launch_nths_nt<6>(a, st);
        case 9: return launch_nths_nt<9>(a, st);
        case 12: return launch_nths_nt<12>(a, st);
        case 16: return launch_nths_nt<16>(a, st);
        case 18: return launch_nths_nt<18>(a, st);
        default: return HGV_EUNSUPPORTED;
    }
}

// Both parts in one launch; every part streams all row tiles of A (share of the compute units and pacing as for the
// bf16 kernel: the cost of a row tile grows slowly with the part's width).
template <int NT0, int NT1>
int launch_nth_pair(NtHArgs a, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt16h_pair<NT0, NT1>;
    const int rc = raise_lds_h(kern);
    if (rc != HGV_OK) return rc;
    const int cus = cu_count_h();
    int64_t wpb = hgv_cdiv(a.ntiles, cus / 2 > 0 ? cus / 2 : 1);
    if (wpb < nth_min_waves()) wpb = nth_min_waves();
    if (wpb > nth_waves<NT0>()) wpb = nth_waves<NT0>();
    int share0 = cus / 2;
    const int64_t a_bytes = a.M * 4 * (a.lda1 + (a.A2 ? a.lda2 : 0));
    const bool paced = a_bytes >= PACE_MIN_BYTES && cus % 16 == 0 && cus / 2 <= PACE_WGS && wpb <= PACE_WAVES &&
                       hgv_cdiv(a.ntiles, wpb) >= cus / 2 && a.ntiles < ((int64_t)1 << 31);
    a.paced = paced ? 1 : 0;
    if (!paced) {
        int64_t best = -1;
        const int lo = cus / 4 > 0 ? cus / 4 : 1;
        for (int s0 = lo; s0 <= cus - lo && s0 < cus; ++s0) {
            const int64_t r0 = hgv_cdiv(a.ntiles, (int64_t)s0 * wpb), r1 = hgv_cdiv(a.ntiles, (int64_t)(cus - s0) * wpb);
            const int64_t c0 = r0 * (11 + NT0), c1 = r1 * (11 + NT1);
            const int64_t c = c0 > c1 ? c0 : c1;
            const int d = s0 > cus / 2 ? s0 - cus / 2 : cus / 2 - s0, dbest = share0 > cus / 2 ? share0 - cus / 2 : cus / 2 - share0;
            if (best < 0 || c < best || (c == best && d < dbest)) { best = c; share0 = s0; }
        }
    }
    const int share1 = cus - share0;
    const int64_t want = hgv_cdiv(a.ntiles, wpb);
    const int wgs0 = (int)(want < share0 ? want : share0), wgs1 = (int)(want < share1 ? want : share1);
    a.wgs0 = wgs0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(wgs0 + wgs1)), dim3((unsigned)(64 * wpb)), lds, st, a);
    return HGV_OK;
}

int launch_nth_both(const NtHArgs& a, int nt0, int nt1, size_t lds, hipStream_t st) {
#define HGV_PAIR(N0)                                                                                   \
    case N0: return nt1 == N0 ? launch_nth_pair<N0, N0>(a, lds, st) : launch_nth_pair<N0, (N0 > 1 ? N0 - 1 : 1)>(a, lds, st);
    switch (nt0) {
        HGV_PAIR(1) HGV_PAIR(2) HGV_PAIR(3) HGV_PAIR(4) HGV_PAIR(5) HGV_PAIR(6) HGV_PAIR(7) HGV_PAIR(8) HGV_PAIR(9)
        default: return HGV_EUNSUPPORTED;
    }
#undef HGV_PAIR
}

}  // namespace

// 1 when hgv_gemm_nt16h can run this shape
extern "C" int hgv_gemm_nt16h_supported(int N, int K1, int K2) {
    GeoH g;
    return (N > 0 && K1 > 0 && K2 >= 0 && geo_h(N, K1, K2, &g)) ? 1 : 0;
}

// image (two fp16 pieces per element) followed by one inverse scale per image row
extern "C" size_t hgv_pack16h_bytes(int N, int K1, int K2) {
    GeoH g;
    if (!(N > 0 && K1 > 0 && K2 >= 0 && geo_h(N, K1, K2, &g))) return 0;
    return hgv_align256(part_bytes_h(g.rows, g.chunks) * g.parts + (size_t)g.parts * g.tpp * 16 * 4);
}

// Up to 24 images in one launch: job q packs [W1[q] | W2[q]] (W2[q] may be null, then K2[q] = 0) into out[q].
extern "C" int hgv_pack16h_weights_multi(int njobs, const float* const* W1, const int64_t* sn1, const int64_t* sk1, const int* K1,
                                         const float* const* W2, const int64_t* sn2, const int64_t* sk2, const int* K2,
                                         const int* N, void* const* out, hipStream_t stream) {
    HGV_REQUIRE(njobs >= 1 && njobs <= PACKH_MAX_JOBS && W1 && sn1 && sk1 && K1 && W2 && sn2 && sk2 && K2 && N && out, HGV_EINVAL,
                "pack16h_weights_multi: 1..%d jobs and non-null arrays expected", PACKH_MAX_JOBS);
    PackHJobs jobs{};
    jobs.njobs = njobs;
    int blocks = 0;
    for (int q = 0; q < njobs; ++q) {
        PackHArgs& a = jobs.job[q];
        HGV_REQUIRE(W1[q] && out[q] && N[q] > 0 && K1[q] > 0 && (W2[q] ? K2[q] > 0 : K2[q] == 0), HGV_EINVAL, "pack16h_weights: bad arguments");
        HGV_REQUIRE(geo_h(N[q], K1[q], K2[q], &a.g), HGV_EUNSUPPORTED, "pack16h_weights: %d x (%d + %d) does not fit in LDS", N[q], K1[q], K2[q]);
        HGV_REQUIRE(hgv_aligned(out[q], 16), HGV_EINVAL, "pack16h_weights: output must be 16-byte aligned");
        a.W1 = W1[q]; a.sn1 = sn1[q]; a.sk1 = sk1[q]; a.W2 = W2[q]; a.sn2 = sn2[q]; a.sk2 = sk2[q]; a.out = (_Float16*)out[q];
        a.cs1 = a.ct1 = a.cs2 = a.ct2 = a.bias_in = nullptr; a.bias_out = nullptr; a.ind1 = 0;
        jobs.first_block[q] = blocks;
        blocks += (a.g.parts * a.g.tpp * 16 + 3) / 4;
    }
    for (int q = njobs; q <= PACKH_MAX_JOBS; ++q) jobs.first_block[q] = blocks;
    hipLaunchKernelGGL(k_pack16h, dim3((unsigned)blocks), dim3(256), 0, stream, jobs);
    HGV_CHECK_LAUNCH("pack16h_weights");
    return HGV_OK;
}

// One image of [W1 diag(cs1) | W2 diag(cs2)] (cs: per-column factors, nullable = ones) and the bias that absorbs the
// column shifts: bias_out[n] = bias_in[n] + sum_k W1[n,k] ct1[k] + sum_k W2[n,k] ct2[k] (bias_in / ct nullable = zeros).
// For operands that are handed over un-normalised: with x_s = a_s * cs_s + ct_s column by column (a batch-norm output
// kept as its input a and the per-column scale / shift),  x_1 W1^T + x_2 W2^T + bias = a_1 (W1 diag(cs1))^T + a_2 (W2
// diag(cs2))^T + bias_out.  W2 nullable (then K2 = 0).  cs / ct hold K1 / K2 floats.
// indicator1 != 0: operand 1 carries one more column than W1 has (K1 = columns of W1 + 1), an indicator in {0, 1} per
// row (hgv_agg_fwd_ind writes it: rows without neighbours aggregate to 0, not to the shift); the image's column K1 - 1
// is sum_k W1[n,k] ct1[k], which then stays out of bias_out.
extern "C" int hgv_pack16h_weights_affine(const float* W1, int64_t sn1, int64_t sk1, int K1, const float* cs1, const float* ct1,
                                          int indicator1, const float* W2, int64_t sn2, int64_t sk2, int K2, const float* cs2,
                                          const float* ct2, int N, const float* bias_in, float* bias_out, void* out,
                                          hipStream_t stream) {
    HGV_REQUIRE(W1 && out && bias_out && N > 0 && K1 > 0 && (W2 ? K2 > 0 : K2 == 0), HGV_EINVAL, "pack16h_weights_affine: bad arguments");
    HGV_REQUIRE(!indicator1 || (K1 > 1 && ct1), HGV_EINVAL, "pack16h_weights_affine: an indicator column needs a shift vector and K1 > 1");
    PackHJobs jobs{};
    jobs.njobs = 1;
    PackHArgs& a = jobs.job[0];
    HGV_REQUIRE(geo_h(N, K1, K2, &a.g), HGV_EUNSUPPORTED, "pack16h_weights_affine: %d x (%d + %d) does not fit in LDS", N, K1, K2);
    HGV_REQUIRE(hgv_aligned(out, 16), HGV_EINVAL, "pack16h_weights_affine: output must be 16-byte aligned");
    a.W1 = W1; a.sn1 = sn1; a.sk1 = sk1; a.W2 = W2; a.sn2 = sn2; a.sk2 = sk2; a.out = (_Float16*)out;
    a.cs1 = cs1; a.ct1 = ct1; a.cs2 = W2 ? cs2 : nullptr; a.ct2 = W2 ? ct2 : nullptr; a.bias_in = bias_in; a.bias_out = bias_out;
    a.ind1 = indicator1 ? 1 : 0;
    jobs.first_block[0] = 0;
    const int blocks = (a.g.parts * a.g.tpp * 16 + 3) / 4;
    for (int q = 1; q <= PACKH_MAX_JOBS; ++q) jobs.first_block[q] = blocks;
    hipLaunchKernelGGL(k_pack16h, dim3((unsigned)blocks), dim3(256), 0, stream, jobs);
    HGV_CHECK_LAUNCH("pack16h_weights_affine");
    return HGV_OK;
}

// C[M,N] = A1 W1^T (+ A2 W2^T) (+ bias) from the scaled fp16 image.  Buffer rules as for hgv_gemm_nt16.
struct NtHPair { float* C2; const uint32_t* bits2; float* partial2; };
struct NtHBwd { const float* d; int64_t ldd; const float* gamma; const float* mean; const float* rstd; const float* sums; const float* add; int64_t ldadd; };
static int gemm_nt16h_any(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                          const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int gate_run_if,
                          hipStream_t stream, const uint32_t* act_bits = nullptr, int64_t act_ldb = 0, int act_relu = 0,
                          float act_scale = 1.f, float* stat_partial = nullptr, const NtHBwd* bwd = nullptr, const NtHPair* pair = nullptr) {
    HGV_REQUIRE(M >= 0 && N > 0 && K1 > 0, HGV_EINVAL, "gemm_nt16h: bad sizes");
    if (M == 0) return HGV_OK;
    NtHArgs a{};
    HGV_REQUIRE(geo_h(N, K1, A2 ? K2 : 0, &a.g), HGV_EUNSUPPORTED, "gemm_nt16h: %d x (%d + %d) does not fit in LDS", N, K1, K2);
    HGV_REQUIRE(A1 && hgv_aligned(A1, 16) && lda1 % 4 == 0 && lda1 >= (K1 + 3) / 4 * 4 && Wp && hgv_aligned(Wp, 16), HGV_EINVAL,
                "gemm_nt16h: bad/misaligned buffer (source 1)");
    if (A2) HGV_REQUIRE(K2 > 0 && hgv_aligned(A2, 16) && lda2 % 4 == 0 && lda2 >= (K2 + 3) / 4 * 4, HGV_EINVAL,
                        "gemm_nt16h: bad/misaligned buffer (source 2)");
    HGV_REQUIRE(C && hgv_aligned(C, 16) && ldc % 4 == 0 && ldc >= N, HGV_EINVAL, "gemm_nt16h: bad/misaligned output");
    a.A1 = A1; a.lda1 = lda1; a.A2 = A2; a.lda2 = lda2; a.Wp = (const _Float16*)Wp; a.bias = bias;
    a.C = C; a.ldc = ldc; a.M = M; a.ntiles = hgv_cdiv(M, 16);
    a.gate = gate; a.gate_run_if = gate_run_if;
    a.pace_slot = hgv_pace_slot(stream);
    a.act_bits = act_bits; a.act_ldb = act_ldb; a.act_relu = act_relu; a.act_scale = act_scale;
    a.stat_partial = stat_partial; a.stat_hp = (N + 3) / 4 * 4;
    if (act_bits) HGV_REQUIRE(!a.g.stream && a.g.parts == 1 && a.g.tiles == 9 && act_ldb >= 5, HGV_EUNSUPPORTED,
                              "gemm_nt16h_act: only one-part images of 9 column tiles (129..144 outputs) store activated");
    int rc;
    const NtrPlan plan = ntr_plan(a.g, M, A2 != nullptr, act_bits != nullptr, pair != nullptr);
    if (pair) {
        HGV_REQUIRE(act_bits && stat_partial && pair->C2 && pair->bits2 && pair->partial2 && hgv_aligned(pair->C2, 16) && !bwd && !gate,
                    HGV_EINVAL, "gemm_nt16h_act2_stats: null / misaligned second output, keep bits or partial buffer");
        HGV_REQUIRE(plan.ok, HGV_EUNSUPPORTED, "gemm_nt16h_act2_stats: (%d + %d) -> %d is not the shape of the two-draw store", K1, K2, N);
        a.C2 = pair->C2; a.act_bits2 = pair->bits2; a.stat_partial2 = pair->partial2;
    }
    if (bwd) {
        HGV_REQUIRE(!act_bits && !bias && !gate && ntr_bwd_shape(a.g, plan, A2 != nullptr), HGV_EUNSUPPORTED,
                    "gemm_nt16h_bwd: shape (%d + %d) -> %d is not one of the fused ones", K1, K2, N);
        HGV_REQUIRE(bwd->d && hgv_aligned(bwd->d, 16) && bwd->ldd % 4 == 0 && bwd->ldd >= ldc && bwd->mean &&
                    bwd->rstd && bwd->sums && ldc <= 144, HGV_EINVAL, "gemm_nt16h_bwd: bad/misaligned level buffers");
        a.bwd_d = bwd->d; a.bwd_ldd = bwd->ldd; a.bwd_gamma = bwd->gamma; a.bwd_mean = bwd->mean; a.bwd_rstd = bwd->rstd; a.bwd_sums = bwd->sums;
        if (bwd->add) {
            HGV_REQUIRE(hgv_aligned(bwd->add, 16) && bwd->ldadd % 4 == 0 && bwd->ldadd >= ldc && ((A2 && a.g.chunks == 9) || (!A2 && a.g.chunks == 5)),
                        HGV_EUNSUPPORTED, "gemm_nt16h_bwd: an addend goes with 129..144 (+ 129..144) source columns only");
            a.bwd_add = bwd->add; a.bwd_ldadd = bwd->ldadd;
        }
    }
    if (plan.ok) {
        rc = launch_ntr(a, plan, stream);
    } else if (a.g.stream) {
        rc = launch_nths(a, stream);
    } else {
        const size_t lds = part_bytes_h(a.g.rows, a.g.chunks) + side_bytes_h(a.g.chunks, 9);
        if (a.g.parts == 1) {
            rc = launch_nth_part(a, 0, a.g.tiles, lds, stream);
        } else {
            const int nt1 = a.g.tiles - a.g.tpp;
            HGV_REQUIRE(nt1 >= 1 && (nt1 == a.g.tpp || nt1 == a.g.tpp - 1), HGV_EUNSUPPORTED, "gemm_nt16h: unexpected split %d + %d", a.g.tpp, nt1);
            rc = launch_nth_both(a, a.g.tpp, nt1, lds, stream);
        }
    }
    if (rc != HGV_OK) return rc;
    HGV_CHECK_LAUNCH("gemm_nt16h");
    return HGV_OK;
}

extern "C" int hgv_gemm_nt16h(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                              const float* bias, int64_t M, int N, float* C, int64_t ldc, hipStream_t stream) {
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, nullptr, 0, stream);
}

// 1 when hgv_gemm_nt16h_act can run this shape
extern "C" int hgv_gemm_nt16h_act_supported(int N, int K1, int K2) {
    GeoH g;
    return (N > 0 && K1 > 0 && K2 >= 0 && geo_h(N, K1, K2, &g) && !g.stream && g.parts == 1 && g.tiles == 9) ? 1 : 0;
}

// hgv_gemm_nt16h that stores dropout(relu(C)) (relu optional): keep bit of element (row, c) = bit c & 31 of
// keep_bits[row * ld_bits + (c >> 5)] (hgv_dropout_bits), kept values times scale = 1 / (1 - p).  A dropped element is
// stored as -0.0f when relu == 0 (a kept zero as +0.0f), as hgv_bn_stats_act leaves them.
extern "C" int hgv_gemm_nt16h_act(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                  const float* bias, int64_t M, int N, float* C, int64_t ldc, const uint32_t* keep_bits,
                                  int64_t ld_bits, int relu, float scale, hipStream_t stream) {
    HGV_REQUIRE(keep_bits != nullptr, HGV_EINVAL, "gemm_nt16h_act: null keep bits");
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, nullptr, 0, stream, keep_bits, ld_bits, relu ? 1 : 0, scale);
}

// hgv_gemm_nt16h_act that also leaves the column statistics of what it stored: stat_partial[parts][2][Hp] (Hp = 4 *
// ceil(N / 4)) receives per part p of hgv_gemm_nt16h_act_parts(M, N, K1, K2) the (mean, M2) of the rows of the 16-row groups p,
// p + parts, p + 2 parts, ... - the layout hgv_bn_stats_merge finishes.  Plain fp32 sums of d and d^2 per part: meant for
// levels behind a dropout of rate >= 0.2, whose zeros bound mean^2 / variance by (1 - p) / p.
extern "C" int hgv_gemm_nt16h_act_parts(int64_t M, int N, int K1, int K2) {
    GeoH g;
    if (M < 1) M = 1;
    if (N > 0 && K1 > 0 && K2 >= 0 && geo_h(N, K1, K2, &g) && ntr_plan(g, M, K2 > 0, true).ok) return (int)ntr_grid(M);
    int64_t wpb, blocks;
    nth_grid<9>(hgv_cdiv(M, 16), &wpb, &blocks);
    return (int)blocks;
}

extern "C" int hgv_gemm_nt16h_act_stats(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                        const float* bias, int64_t M, int N, float* C, int64_t ldc, const uint32_t* keep_bits,
                                        int64_t ld_bits, int relu, float scale, float* stat_partial, hipStream_t stream) {
    HGV_REQUIRE(keep_bits != nullptr && stat_partial != nullptr, HGV_EINVAL, "gemm_nt16h_act_stats: null keep bits / partial buffer");
    HGV_REQUIRE(M >= 2, HGV_EINVAL, "gemm_nt16h_act_stats: statistics need at least two rows");
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, nullptr, 0, stream, keep_bits, ld_bits, relu ? 1 : 0, scale,
                          stat_partial);
}

// 1 when hgv_gemm_nt16h_bwd can run this shape at M rows (addend != 0: with an addend)
extern "C" int hgv_gemm_nt16h_bwd_supported(int64_t M, int N, int K1, int K2, int addend) {
    GeoH g;
    if (!(M > 0 && N > 0 && K1 > 0 && K2 >= 0 && geo_h(N, K1, K2, &g))) return 0;
    if (addend && g.chunks != (K2 > 0 ? 9 : 5)) return 0;
    return ntr_bwd_shape(g, ntr_plan(g, M, K2 > 0, false), K2 > 0) ? 1 : 0;
}

// The grad-input product of a consumer whose operand was a level relu? -> Dropout -> BatchNorm1d handed over as its stored input
// d [M, N] and per-column vectors: instead of g = A1 W1^T (+ A2 W2^T), the gradient with respect to the level's OUTPUT, it stores
//   dy = k * gamma rstd (g - sums[0] / M - xhat sums[1] / M),  xhat = (d - mean) rstd,  k = 1 / (1 - p) where d was kept, else 0
// (relu: kept = d > 0; otherwise dropped elements are the negative zeros) - the gradient with respect to the level's
// pre-activation, what hgv_bn_act_bwd_pre_alg makes of g in a pass of its own.  sums[2][4 ceil(N / 4)]: column sums of g and of
// g xhat over all M rows (hgv_bn_bwd_sums_alg); gamma nullable (ones).  dy needs ldc <= 144 and receives zeros in its pad columns.
// addend (nullable, [M, N] with leading dimension ldadd): g = A1 W1^T (+ A2 W2^T) + addend - the share of the level's incoming
// gradient that its other consumer has already written; only with one source of 129..160 columns or two of 129..144 each.
extern "C" int hgv_gemm_nt16h_bwd(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                  int64_t M, int N, float* dy, int64_t lddy, const float* d, int64_t ldd, int relu, float p,
                                  const float* gamma, const float* mean, const float* rstd, const float* sums, const float* addend,
                                  int64_t ldadd, hipStream_t stream) {
    HGV_REQUIRE(p >= 0.f && p < 1.f, HGV_EINVAL, "gemm_nt16h_bwd: dropout p=%f outside [0,1)", p);
    const NtHBwd b{d, ldd, gamma, mean, rstd, sums, addend, ldadd};
    const float scale = 1.0f / (float)(1.0 - (double)p);          // as the level's own kernels form it
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, nullptr, M, N, dy, lddy, nullptr, 0, stream, nullptr, 0, relu ? 1 : 0, scale,
                          nullptr, &b);
}

// hgv_gemm_nt16h_act_stats for two dropout draws of the one product: C / stat_partial take the draw of keep_bits, C2 /
// stat_partial2 the draw of keep_bits2 (same leading dimensions; hgv_gemm_nt16h_act2_parts(M) parts each).  Built for the
// first convolution of the two encode passes: two sources of 97...112 columns each, 129...144 outputs
// (hgv_gemm_nt16h_act2_supported).
extern "C" int hgv_gemm_nt16h_act2_parts(int64_t M) { return (int)ntr_grid(M < 1 ? 1 : M); }

extern "C" int hgv_gemm_nt16h_act2_supported(int64_t M, int N, int K1, int K2) {
    GeoH g;
    if (!(M > 1 && N > 0 && K1 > 0 && K2 > 0 && geo_h(N, K1, K2, &g)) || g.stream) return 0;
    return ntr_plan(g, M, true, true, true).ok ? 1 : 0;
}

extern "C" int hgv_gemm_nt16h_act2_stats(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                         const float* bias, int64_t M, int N, float* C, float* C2, int64_t ldc,
                                         const uint32_t* keep_bits, const uint32_t* keep_bits2, int64_t ld_bits, int relu, float scale,
                                         float* stat_partial, float* stat_partial2, hipStream_t stream) {
    HGV_REQUIRE(M >= 2, HGV_EINVAL, "gemm_nt16h_act2_stats: statistics need at least two rows");
    HGV_REQUIRE(A2 != nullptr && keep_bits && stat_partial, HGV_EINVAL, "gemm_nt16h_act2_stats: two sources, keep bits and partial buffers expected");
    const NtHPair pr{C2, keep_bits2, stat_partial2};
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, nullptr, 0, stream, keep_bits, ld_bits, relu ? 1 : 0, scale,
                          stat_partial, nullptr, &pr);
}

// hgv_gemm_nt16h that does its work only when (*gate != 0) == run_if_set (gate: device int written earlier on the
// stream, e.g. by hgv_row_range_flag); otherwise every workgroup returns at once and C is left untouched.
extern "C" int hgv_gemm_nt16h_gated(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                    const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int run_if_set,
                                    hipStream_t stream) {
    HGV_REQUIRE(gate != nullptr, HGV_EINVAL, "gemm_nt16h_gated: null gate");
    return gemm_nt16h_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, gate, run_if_set ? 1 : 0, stream);
}

// ---- range guard of the two-piece fp16 split ------------------------------------------------------------------------
// A row is scaled by one power of two taken from its largest entry; an entry more than 2^16 below it reaches the fp16
// pieces as a subnormal and keeps an absolute, not a relative, accuracy.  This pass decides, per launch, whether any
// row of [A1 | A2] holds a non-zero entry more than 2^log2_limit below the row's largest one: flag[0] = 1 if so, else 0.
namespace {
__global__ __launch_bounds__(256) void k_row_range(const float* __restrict__ A1, int64_t lda1, int c1, const float* __restrict__ A2,
                                                   int64_t lda2, int c2, int64_t M, float limit, int* __restrict__ flag) {
    // 16 lanes per row, 16-byte chunks; exponent arithmetic on the bit patterns (zeros and denormals count as zero)
    const int sub = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    float amax = 0.f, amin = 3.0e38f;
    if (row < M) {
        for (int src = 0; src < 2; ++src) {
            const float* base = src ? A2 : A1;
            if (!base) continue;
            const f32x4* p = reinterpret_cast<const f32x4*>(base + row * (src ? lda2 : lda1));
            const int nch = src ? c2 : c1;
            for (int c = sub; c < nch; c += 16) {
                const f32x4 v = p[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = __builtin_fabsf(v[r]);
                    amax = fmaxf(amax, x);
                    amin = x >= 1.17549435e-38f ? fminf(amin, x) : amin;
                }
            }
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, d));
        amin = fminf(amin, __shfl_xor(amin, d));
    }
    const bool wide = row < M && amin < 3.0e38f && amax > amin * limit;
    if (__builtin_amdgcn_ballot_w64(wide) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace

extern "C" int hgv_row_range_flag(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, int64_t M,
                                  int log2_limit, int* flag, hipStream_t stream) {
    HGV_REQUIRE(M >= 0 && K1 > 0 && A1 && flag && log2_limit >= 1 && log2_limit <= 60, HGV_EINVAL, "row_range_flag: bad arguments");
    HGV_REQUIRE(hgv_aligned(A1, 16) && lda1 % 4 == 0 && lda1 >= (K1 + 3) / 4 * 4, HGV_EINVAL, "row_range_flag: bad/misaligned buffer (source 1)");
    if (A2) HGV_REQUIRE(K2 > 0 && hgv_aligned(A2, 16) && lda2 % 4 == 0 && lda2 >= (K2 + 3) / 4 * 4, HGV_EINVAL,
                        "row_range_flag: bad/misaligned buffer (source 2)");
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), stream);
    HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "row_range_flag: memset failed: %s", hipGetErrorString(e));
    if (M == 0) return HGV_OK;
    // whole 16-byte chunks are read: pad columns must be zero (they are ignored as zeros)
    hipLaunchKernelGGL(k_row_range, dim3((unsigned)hgv_cdiv(M * 16, 256)), dim3(256), 0, stream, A1, lda1, (K1 + 3) / 4,
                       A2, lda2, A2 ? (K2 + 3) / 4 : 0, M, __builtin_ldexpf(1.f, log2_limit), flag);
    HGV_CHECK_LAUNCH("row_range_flag");
    return HGV_OK;
}
