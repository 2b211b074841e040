// Projections at fp32 accuracy on the bf16 matrix cores ("bf16x6").
//
// fp32 MFMAs run on the vector ALUs and are power-limited to ~115-125 TF/s with live data on
// MI355X; the bf16 matrix cores are 16x faster per instruction.  Every fp32 operand is split into
// three bf16 pieces x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 significant
// bits in total) and the product a*w is formed from the six significant piece products
//   al*wh + ah*wl + am*wm + ah*wm + am*wh + ah*wh        (dropped terms are < 2^-24 |a w|)
// each by one v_mfma_f32_16x16x32_bf16 with fp32 accumulation: measured maximum and rms error
// against fp64 are the same as those of the fp32 MFMA chain (piece products are exact in fp32).
//
// Same operator contract as the fp32 kernels (gvae/gvae_skipcon.py:61-100):
//   C[M,N] = A1[M,K1] W1^T (+ A2[M,K2] W2^T) (+ bias)
// Weights are split and packed once per use by k_pack16 into the layout the kernel keeps in LDS:
//   [part][chunk c][piece s][row r][32 k], k-contiguous rows of 64 bytes whose four 16-byte pieces
//   are XOR-swizzled with (r & 8 ? 3 : 0) so that a wave's ds_read_b128 is conflict-free.
// The k axis is the concatenation of the two sources in blocks of 8 columns (each source padded
// to a multiple of 8, the total to a multiple of 32); a "part" is a range of <= NT n tiles whose
// image fits the 160 KB of LDS - a weight that does not fit whole is split along N and the rows of
// A are streamed once per part.  A workgroup keeps its part resident, then every wave streams 16-row
// tiles on its own (no barrier in the main loop): it loads 8 consecutive floats per lane and step,
// splits them in registers (VALU work, which overlaps with the matrix cores) and reads the weight
// pieces of each n tile from LDS.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr size_t LDS16_MAX = 160 * 1024;

struct Geo16 {
    int N, K1, K2;
    int nb1, nb2;        // 8-column blocks per source
    int chunks;          // 32-wide k chunks over the concatenation
    int tiles;           // n tiles (16 columns each)
    int parts, tpp;      // parts along N, n tiles per part (last part may hold fewer)
};

__host__ __device__ inline size_t part_bytes16(int tiles_in_part, int chunks) { return (size_t)chunks * 3 * tiles_in_part * 16 * 64; }

inline bool geo16(int N, int K1, int K2, Geo16* g) {
    g->N = N; g->K1 = K1; g->K2 = K2;
    g->nb1 = (K1 + 7) / 8; g->nb2 = (K2 + 7) / 8;
    g->chunks = (g->nb1 + g->nb2 + 3) / 4;
    g->tiles = (N + 15) / 16;
    // table (chunks x 4 lanes groups x 16 B) + bias row live in LDS next to the image
    for (int parts = 1; parts <= 2; ++parts) {
        const int tpp = (g->tiles + parts - 1) / parts;
        if (tpp <= 9 && part_bytes16(tpp, g->chunks) + (size_t)g->chunks * 64 + 9 * 16 * 4 + 256 <= LDS16_MAX) {
            g->parts = parts; g->tpp = tpp;
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

struct Pack16Args {
    const float* W1; int64_t sn1, sk1; const float* W2; int64_t sn2, sk2;
    Geo16 g;
    __bf16* out;
};

constexpr int PACK16_MAX_JOBS = 4;
struct Pack16Jobs {
    Pack16Args job[PACK16_MAX_JOBS];
    int first_block[PACK16_MAX_JOBS + 1];    // job q owns blocks [first_block[q], first_block[q+1])
    int njobs;
};

// one thread per (part, chunk, row, k) element of an image; writes the three pieces.  Up to four images per launch.
__global__ __launch_bounds__(256) void k_pack16(const Pack16Jobs jobs) {
    int q = 0;
#pragma unroll
    for (int j = 1; j < PACK16_MAX_JOBS; ++j)
        if (j < jobs.njobs && (int)blockIdx.x >= jobs.first_block[j]) q = j;
    const Pack16Args& a = jobs.job[q];
    const Geo16& g = a.g;
    const int64_t per_part = (int64_t)g.chunks * g.tpp * 16 * 32;
    const int64_t total = per_part * g.parts;
    {
        const int64_t e = (int64_t)((int)blockIdx.x - jobs.first_block[q]) * blockDim.x + threadIdx.x;
        if (e >= total) return;
        const int part = (int)(e / per_part);
        int64_t l = e - (int64_t)part * per_part;
        const int rows = g.tpp * 16;
        const int c = (int)(l / (rows * 32)); l -= (int64_t)c * rows * 32;
        const int r = (int)(l >> 5), kk = (int)(l & 31);
        const int n = part * rows + r;
        const int blk = 4 * c + (kk >> 3), j = kk & 7;
        float x = 0.f;
        if (n < g.N) {
            if (blk < g.nb1) { const int k = 8 * blk + j; if (k < g.K1) x = a.W1[(int64_t)n * a.sn1 + (int64_t)k * a.sk1]; }
            else if (blk < g.nb1 + g.nb2) { const int k = 8 * (blk - g.nb1) + j; if (k < g.K2) x = a.W2[(int64_t)n * a.sn2 + (int64_t)k * a.sk2]; }
        }
        __bf16 h, m, lo;
        split3(x, h, m, lo);
        const int piece = (kk >> 3) ^ ((r & 8) ? 3 : 0);
        const int64_t base = (int64_t)part * g.chunks * 3 * rows * 32 + (int64_t)c * 3 * rows * 32 + (int64_t)r * 32 + piece * 8 + j;
        a.out[base] = h;
        a.out[base + (int64_t)rows * 32] = m;
        a.out[base + (int64_t)2 * rows * 32] = lo;
    }
}

int fill_pack16_job(Pack16Args& a, const float* W1, int64_t sn1, int64_t sk1, int K1, const float* W2, int64_t sn2, int64_t sk2,
                    int K2, int N, void* out) {
    HGV_REQUIRE(W1 && out && N > 0 && K1 > 0 && (W2 ? K2 > 0 : K2 == 0), HGV_EINVAL, "pack16_weights: bad arguments");
    HGV_REQUIRE(geo16(N, K1, K2, &a.g), HGV_EUNSUPPORTED, "pack16_weights: %d x (%d + %d) does not fit in LDS", N, K1, K2);
    HGV_REQUIRE(hgv_aligned(out, 16), HGV_EINVAL, "pack16_weights: output must be 16-byte aligned");
    a.W1 = W1; a.sn1 = sn1; a.sk1 = sk1; a.W2 = W2; a.sn2 = sn2; a.sk2 = sk2; a.out = (__bf16*)out;
    return HGV_OK;
}

int launch_pack16(Pack16Jobs& jobs, hipStream_t stream) {
    int blocks = 0;
    for (int q = 0; q < jobs.njobs; ++q) {
        const Geo16& g = jobs.job[q].g;
        jobs.first_block[q] = blocks;
        blocks += (int)hgv_cdiv((int64_t)g.parts * g.chunks * g.tpp * 16 * 32, 256);
    }
    for (int q = jobs.njobs; q <= PACK16_MAX_JOBS; ++q) jobs.first_block[q] = blocks;
    hipLaunchKernelGGL(k_pack16, dim3((unsigned)blocks), dim3(256), 0, stream, jobs);
    HGV_CHECK_LAUNCH("pack16_weights");
    return HGV_OK;
}

struct Nt16Args {
    const float* A1; int64_t lda1; const float* A2; int64_t lda2;
    Geo16 g;
    const __bf16* Wp; const float* bias;
    float* C; int64_t ldc; int64_t M;
    int64_t ntiles;            // 16-row tiles
    int wgs0;                  // two-part launches: workgroups [0, wgs0) work on part 0, the rest on part 1
};

// waves per workgroup: 16 while the registers allow 4 waves per SIMD (<= 128 VGPRs), 12 for the widest parts
template <int NT> constexpr int nt16_waves() { return NT > 6 ? 12 : 16; }

template <int NT> constexpr int nt16_ring() { return NT > 6 ? 3 : 4; }

// The work of one workgroup: part `part` of the image (exactly NT n tiles) times the row tiles
// wg_rank, wg_rank + wgs, ... (dealt to its waves).
template <int NT, int RING>
__device__ __forceinline__ void nt16_body(const Nt16Args& a, unsigned char* lds_raw, const int part, const int wg_rank, const int wgs) {
    const Geo16& g = a.g;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6), wpb = blockDim.x >> 6;
    const int t0 = part * g.tpp;                                   // first n tile of the part; it holds exactly NT tiles
    const int rows = g.tpp * 16;                                   // rows per piece image (parts are stored with tpp tiles)
    __bf16* Wl = reinterpret_cast<__bf16*>(lds_raw);
    const size_t img = part_bytes16(g.tpp, g.chunks);
    int4* tab = reinterpret_cast<int4*>(lds_raw + img);            // [chunks][4]: byte offsets of the two halves, source
    float* bias_s = reinterpret_cast<float*>(lds_raw + img + (size_t)g.chunks * 64);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)part * img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds_raw);
        // batches of 8 loads in flight per thread: one load -> store per iteration would pay the L2 latency ~10 times
        const int npieces = (int)(img / 16);
        for (int e0 = t; e0 < npieces; e0 += 8 * blockDim.x) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + u * (int)blockDim.x, npieces - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * (int)blockDim.x < npieces) dst[e0 + u * (int)blockDim.x] = v[u];
        }
        for (int e = t; e < g.chunks * 4; e += blockDim.x) {
            const int blk = e;                                     // chunk * 4 + kq
            int second = 0, col = 0; int64_t ld = a.lda1;
            if (blk < g.nb1) { col = 8 * blk; }
            else if (blk < g.nb1 + g.nb2) { second = 1; col = 8 * (blk - g.nb1); ld = a.lda2; }
            else { col = 0; }                                      // padding block: weights are zero there
            // both 16-byte halves stay inside the row's leading dimension (columns past K meet zero weights)
            const int c0 = (int)min((int64_t)col, ld - 4), c1 = (int)min((int64_t)col + 4, ld - 4);
            tab[e] = make_int4(c0 * 4, c1 * 4, second, 0);
        }
        for (int e = t; e < NT * 16; e += blockDim.x) {
            const int n = t0 * 16 + e;
            bias_s[e] = (a.bias && n < g.N) ? a.bias[n] : 0.f;
        }
    }
    __syncthreads();

    const int64_t gw = (int64_t)wib * wgs + wg_rank, nw = (int64_t)wpb * wgs;
    const int64_t my_tiles = gw < a.ntiles ? (a.ntiles - gw + nw - 1) / nw : 0;
    const int steps = g.chunks;

    f32x4 acc[NT];
    // per-lane LDS byte offset of (row i, swizzled piece kq) inside a piece image
    const int wlane = i * 64 + ((kq ^ ((i & 8) ? 3 : 0)) * 16);

    // load cursor
    int64_t ltile = gw, lidx = 0; int lq = 0;
    const char *lp1, *lp2;
    auto row_pointers = [&]() __attribute__((always_inline)) {
        const int64_t row = min(ltile * 16 + i, a.M - 1);
        lp1 = reinterpret_cast<const char*>(a.A1 + row * a.lda1);
        lp2 = reinterpret_cast<const char*>((a.A2 ? a.A2 : a.A1) + row * (a.A2 ? a.lda2 : a.lda1));
    };
    f32x4 ring[RING][2];
    int4 e_next = tab[kq];                   // table entry of the load cursor, read one step early
    auto issue = [&](int d) __attribute__((always_inline)) {
        const int4 e = e_next;
        const char* p = e.z ? lp2 : lp1;
        ring[d][0] = *reinterpret_cast<const f32x4*>(p + e.x);
        ring[d][1] = *reinterpret_cast<const f32x4*>(p + e.y);
        if (++lq == steps) {
            lq = 0;
            if (lidx + 1 < my_tiles) { ++lidx; ltile += nw; row_pointers(); }     // past the end: the last tile again
        }
        e_next = tab[lq * 4 + kq];
    };
    // n tiles are worked on in groups of GS: the six piece products of a group are issued term by term
    // across its tiles, so MFMAs that depend on each other are GS instructions apart.  The order of
    // the MFMAs is pinned (other instructions may move across the fences): left alone, the scheduler
    // strings the six products of one tile together and every one of them waits for its predecessor.
    constexpr int GS = (NT % 3 == 0 || NT > 4) ? 3 : (NT > 1 ? 2 : 1);
    // The split of the NEXT step's A values is computed while this step's MFMAs are in flight: a
    // bf16 MFMA keeps the matrix core busy for 16 cycles but issues in 4, so two VALU instructions fit
    // behind every MFMA (the group barriers below ask for exactly that interleave).
    struct Split { bf16x8 h, m, l; };
    auto split8 = [&](const f32x4 (&av)[2]) __attribute__((always_inline)) {
        Split o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 h, m, l;
            split3(av[j >> 2][j & 3], h, m, l);
            o.h[j] = h; o.m[j] = m; o.l[j] = l;
        }
        return o;
    };
    auto compute = [&](const Split& cur, const f32x4 (&av_next)[2], int c) __attribute__((always_inline)) {
        const unsigned char* base = lds_raw + (size_t)c * 3 * rows * 64 + wlane;
        Split nxt;
        int row_id = 0;      // compile-time after unrolling: which MFMA row of the step we are in
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += GS) {
            bf16x8 wh[GS], wm[GS], wl[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                if (n0 + j < NT) {
                    const unsigned char* w = base + (size_t)(n0 + j) * 16 * 64;
                    wh[j] = *reinterpret_cast<const bf16x8*>(w);
                    wm[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)rows * 64);
                    wl[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)2 * rows * 64);
                }
            }
            // One row = one piece product for the GS tiles of the group (small terms first).  After each
            // of the first 8 rows one element of the next step's operand is split; the fence keeps both
            // where they are written (LDS reads may still move), so the VALU work sits in the shadow
            // of the MFMAs instead of in front of them.
#define HGV_TERM(WP, AP)                                                                                           \
            _Pragma("unroll") for (int j = 0; j < GS; ++j)                                                        \
                if (n0 + j < NT) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WP[j], AP, acc[n0 + j], 0, 0, 0); \
            if (row_id < 8) {                                                                                      \
                __bf16 h_, m_, l_;                                                                                 \
                float x_ = av_next[row_id >> 2][row_id & 3];                                                       \
                asm volatile("" : "+v"(x_));   /* ties the split to this point of the instruction stream */        \
                split3(x_, h_, m_, l_);                                                                            \
                nxt.h[row_id] = h_; nxt.m[row_id] = m_; nxt.l[row_id] = l_;                                        \
            }                                                                                                      \
            ++row_id;                                                                                              \
            __builtin_amdgcn_sched_barrier(0x180);
            HGV_TERM(wl, cur.h) HGV_TERM(wh, cur.l) HGV_TERM(wm, cur.m) HGV_TERM(wm, cur.h) HGV_TERM(wh, cur.m) HGV_TERM(wh, cur.h)
#undef HGV_TERM
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= row_id) {               // fewer than 8 rows in a step (NT = 1): split the rest here
                __bf16 h_, m_, l_;
                split3(av_next[j >> 2][j & 3], h_, m_, l_);
                nxt.h[j] = h_; nxt.m[j] = m_; nxt.l[j] = l_;
            }
        }
        return nxt;
    };

    if (my_tiles > 0) {
        row_pointers();
#pragma unroll
        for (int d = 0; d < RING; ++d) issue(d);
    }
    // Flat loop over all steps of all tiles of this wave.  Every slot issues its load group
    // unconditionally (counted waits); only the last round may hold slots past the end.
    int64_t ctile = gw;
    int cq = 0;
    const int64_t S = my_tiles * steps;
    Split cur;
    if (my_tiles > 0) {
        const f32x4 av0[2] = {ring[0][0], ring[0][1]};
        cur = split8(av0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = *reinterpret_cast<const f32x4*>(&bias_s[16 * n + 4 * kq]);
    }
    for (int64_t u0 = 0; u0 < S; u0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            issue(d);                                            // refills the slot whose values were split one step ago
            if (u0 + d < S) {
                const f32x4 av_next[2] = {ring[(d + 1) % RING][0], ring[(d + 1) % RING][1]};
                cur = compute(cur, av_next, cq);
                if (++cq == steps) {
                    // weights as the row operand: lane (i, kq) register r = C[row i][16 n + 4 kq + r]
                    const int64_t row = ctile * 16 + i;
                    if (row < a.M) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int col = (t0 + n) * 16 + 4 * kq;
                            if (col < a.ldc) *reinterpret_cast<f32x4*>(a.C + row * a.ldc + col) = acc[n];
                        }
                    }
                    cq = 0; ctile += nw;
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n] = *reinterpret_cast<const f32x4*>(&bias_s[16 * n + 4 * kq]);
                }
            }
        }
    }
}

// One part per launch (the whole weight, or one part of a weight split along N).
template <int NT>
__global__ __launch_bounds__(64 * nt16_waves<NT>()) void k_gemm_nt16(const Nt16Args a, const int part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    nt16_body<NT, nt16_ring<NT>()>(a, lds_raw, part, blockIdx.x, gridDim.x);
}

// Both parts of a weight split along N in one launch: the first a.wgs0 workgroups keep part 0
// (NT0 n tiles) resident, the others part 1 (NT1 = NT0 or NT0 - 1 tiles); each group streams all rows of A.
template <int NT0, int NT1>
__global__ __launch_bounds__(64 * nt16_waves<NT0>()) void k_gemm_nt16_pair(const Nt16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if ((int)blockIdx.x < a.wgs0) nt16_body<NT0, nt16_ring<NT0>()>(a, lds_raw, 0, blockIdx.x, a.wgs0);
    else nt16_body<NT1, nt16_ring<NT1>()>(a, lds_raw, 1, (int)blockIdx.x - a.wgs0, (int)gridDim.x - a.wgs0);
}

int cu_count16() {
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
int raise_lds16(Kern kern) {
    static bool raised[64] = {false};          // one instance per kernel type
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)LDS16_MAX);
        HGV_REQUIRE(e == hipSuccess, HGV_ELAUNCH, "gemm_nt16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        raised[dev] = true;
    }
    return HGV_OK;
}

template <int NT>
int launch_nt16(const Nt16Args& a, int part, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt16<NT>;
    const int rc = raise_lds16(kern);
    if (rc != HGV_OK) return rc;
    // one workgroup per compute unit; as many waves (<= 16) as there are tiles to hand out
    const int cus = cu_count16();
    int64_t wpb = hgv_cdiv(a.ntiles, cus);
    if (wpb > nt16_waves<NT>()) wpb = nt16_waves<NT>();
    if (wpb < 1) wpb = 1;
    int64_t blocks = hgv_cdiv(a.ntiles, wpb);
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((unsigned)(64 * wpb)), lds, st, a, part);
    return HGV_OK;
}

int launch_nt16_part(const Nt16Args& a, int part, int nt, size_t lds, hipStream_t st) {
    switch (nt) {
        case 1: return launch_nt16<1>(a, part, lds, st);
        case 2: return launch_nt16<2>(a, part, lds, st);
        case 3: return launch_nt16<3>(a, part, lds, st);
        case 4: return launch_nt16<4>(a, part, lds, st);
        case 5: return launch_nt16<5>(a, part, lds, st);
        case 6: return launch_nt16<6>(a, part, lds, st);
        case 7: return launch_nt16<7>(a, part, lds, st);
        case 8: return launch_nt16<8>(a, part, lds, st);
        default: return launch_nt16<9>(a, part, lds, st);
    }
}

// Both parts in one launch; every part streams all row tiles of A.
template <int NT0, int NT1>
int launch_nt16_pair(Nt16Args a, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt16_pair<NT0, NT1>;
    const int rc = raise_lds16(kern);
    if (rc != HGV_OK) return rc;
    const int cus = cu_count16();
    int64_t wpb = hgv_cdiv(a.ntiles, cus / 2 > 0 ? cus / 2 : 1);
    if (wpb > nt16_waves<NT0>()) wpb = nt16_waves<NT0>();
    if (wpb < 1) wpb = 1;
    // Share of the compute units per part: a wave's time is (rounds of row tiles) x (cost of one row
    // tile), and the cost grows slowly with the part's width (measured: 33 / 36 / 44 us for 4 / 5 / 9
    // n tiles, i.e. ~ 11 + NT), so the split that minimises the slower part is near even and is
    // decided by the rounding of the rounds.  Ties go to the split closest to even.
    int share0 = cus / 2;
    {
        int64_t best = -1;
        for (int s0 = cus / 4 > 0 ? cus / 4 : 1; s0 <= cus - (cus / 4 > 0 ? cus / 4 : 1) && s0 < cus; ++s0) {
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

// n tiles of the two parts are (tpp, tpp) or (tpp, tpp - 1)
int launch_nt16_both(const Nt16Args& a, int nt0, int nt1, size_t lds, hipStream_t st) {
#define HGV_PAIR(N0)                                                                                   \
    case N0: return nt1 == N0 ? launch_nt16_pair<N0, N0>(a, lds, st) : launch_nt16_pair<N0, (N0 > 1 ? N0 - 1 : 1)>(a, lds, st);
    switch (nt0) {
        HGV_PAIR(1) HGV_PAIR(2) HGV_PAIR(3) HGV_PAIR(4) HGV_PAIR(5) HGV_PAIR(6) HGV_PAIR(7) HGV_PAIR(8) HGV_PAIR(9)
        default: return HGV_EUNSUPPORTED;
    }
#undef HGV_PAIR
}

}  // namespace

// 1 when hgv_gemm_nt16 can run this shape (the split weight image, whole or in two parts along N, fits in LDS)
extern "C" int hgv_gemm_nt16_supported(int N, int K1, int K2) {
    Geo16 g;
    return (N > 0 && K1 > 0 && K2 >= 0 && geo16(N, K1, K2, &g)) ? 1 : 0;
}

extern "C" size_t hgv_pack16_bytes(int N, int K1, int K2) {
    Geo16 g;
    if (!(N > 0 && K1 > 0 && K2 >= 0 && geo16(N, K1, K2, &g))) return 0;
    return hgv_align256(part_bytes16(g.tpp, g.chunks) * g.parts);
}

// Split-bf16 image of [W1 | W2] (W2 nullable): element (n,k) of source s at Ws[n*sn_s + k*sk_s].
extern "C" int hgv_pack16_weights(const float* W1, int64_t sn1, int64_t sk1, int K1, const float* W2, int64_t sn2, int64_t sk2,
                                  int K2, int N, void* out, hipStream_t stream) {
    Pack16Jobs jobs{};
    jobs.njobs = 1;
    const int rc = fill_pack16_job(jobs.job[0], W1, sn1, sk1, K1, W2, sn2, sk2, K2, N, out);
    if (rc != HGV_OK) return rc;
    return launch_pack16(jobs, stream);
}

// Up to four images in one launch: job q packs [W1[q] | W2[q]] (W2[q] may be null, then K2[q] = 0) into out[q].
extern "C" int hgv_pack16_weights_multi(int njobs, const float* const* W1, const int64_t* sn1, const int64_t* sk1, const int* K1,
                                        const float* const* W2, const int64_t* sn2, const int64_t* sk2, const int* K2,
                                        const int* N, void* const* out, hipStream_t stream) {
    HGV_REQUIRE(njobs >= 1 && njobs <= PACK16_MAX_JOBS && W1 && sn1 && sk1 && K1 && W2 && sn2 && sk2 && K2 && N && out, HGV_EINVAL,
                "pack16_weights_multi: 1..%d jobs and non-null arrays expected", PACK16_MAX_JOBS);
    Pack16Jobs jobs{};
    jobs.njobs = njobs;
    for (int q = 0; q < njobs; ++q) {
        const int rc = fill_pack16_job(jobs.job[q], W1[q], sn1[q], sk1[q], K1[q], W2[q], sn2[q], sk2[q], K2[q], N[q], out[q]);
        if (rc != HGV_OK) return rc;
    }
    return launch_pack16(jobs, stream);
}

// C[M,N] = A1 W1^T (+ A2 W2^T) (+ bias) from the split-bf16 image.  A1/A2: 16-byte aligned, lda % 4 == 0,
// lda >= ceil(K/4)*4, pad columns finite; C: 16-byte aligned, ldc % 4 == 0; columns [N, ldc) are written as zeros.
extern "C" int hgv_gemm_nt16(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                             const float* bias, int64_t M, int N, float* C, int64_t ldc, hipStream_t stream) {
    HGV_REQUIRE(M >= 0 && N > 0 && K1 > 0, HGV_EINVAL, "gemm_nt16: bad sizes");
    if (M == 0) return HGV_OK;
    Nt16Args a{};
    HGV_REQUIRE(geo16(N, K1, A2 ? K2 : 0, &a.g), HGV_EUNSUPPORTED, "gemm_nt16: %d x (%d + %d) does not fit in LDS", N, K1, K2);
    HGV_REQUIRE(A1 && hgv_aligned(A1, 16) && lda1 % 4 == 0 && lda1 >= (K1 + 3) / 4 * 4 && Wp && hgv_aligned(Wp, 16), HGV_EINVAL,
                "gemm_nt16: bad/misaligned buffer (source 1)");
    if (A2) HGV_REQUIRE(K2 > 0 && hgv_aligned(A2, 16) && lda2 % 4 == 0 && lda2 >= (K2 + 3) / 4 * 4, HGV_EINVAL,
                        "gemm_nt16: bad/misaligned buffer (source 2)");
    HGV_REQUIRE(C && hgv_aligned(C, 16) && ldc % 4 == 0 && ldc >= N, HGV_EINVAL, "gemm_nt16: bad/misaligned output");
    a.A1 = A1; a.lda1 = lda1; a.A2 = A2; a.lda2 = lda2; a.Wp = (const __bf16*)Wp; a.bias = bias;
    a.C = C; a.ldc = ldc; a.M = M; a.ntiles = hgv_cdiv(M, 16);
    const size_t lds = part_bytes16(a.g.tpp, a.g.chunks) + (size_t)a.g.chunks * 64 + 9 * 16 * 4;
    // a part holds exactly its n tiles; the rows of A are streamed once per part
    int rc;
    if (a.g.parts == 1) {
        rc = launch_nt16_part(a, 0, a.g.tiles, lds, stream);
    } else {
        const int nt1 = a.g.tiles - a.g.tpp;
        HGV_REQUIRE(nt1 >= 1 && (nt1 == a.g.tpp || nt1 == a.g.tpp - 1), HGV_EUNSUPPORTED, "gemm_nt16: unexpected split %d + %d", a.g.tpp, nt1);
        rc = launch_nt16_both(a, a.g.tpp, nt1, lds, stream);
    }
    if (rc != HGV_OK) return rc;
    HGV_CHECK_LAUNCH("gemm_nt16");
    return HGV_OK;
}
