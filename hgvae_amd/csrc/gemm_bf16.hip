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
#include <stdlib.h>

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
    int stream;          // 1: the image does not fit in LDS even in two parts; one part of tpp >= tiles n tiles whose
                         //    32-wide k chunks are streamed through LDS (k_gemm_nt16s)
};

__host__ __device__ inline size_t part_bytes16(int tiles_in_part, int chunks) { return (size_t)chunks * 3 * tiles_in_part * 16 * 64; }

inline bool geo16(int N, int K1, int K2, Geo16* g) {
    g->N = N; g->K1 = K1; g->K2 = K2;
    g->nb1 = (K1 + 7) / 8; g->nb2 = (K2 + 7) / 8;
    g->chunks = (g->nb1 + g->nb2 + 3) / 4;
    g->tiles = (N + 15) / 16;
    g->stream = 0;
    // table (chunks x 4 lanes groups x 16 B) + bias row live in LDS next to the image
    for (int parts = 1; parts <= 2; ++parts) {
        const int tpp = (g->tiles + parts - 1) / parts;
        if (tpp <= 9 && part_bytes16(tpp, g->chunks) + (size_t)g->chunks * 64 + 9 * 16 * 4 + 256 <= LDS16_MAX) {
            g->parts = parts; g->tpp = tpp;
            return true;
        }
    }
    // streamed: two chunk buffers of 3 pieces x (16 tpp) rows x 64 B, tpp = instantiated tile count >= tiles
    static const int widths[] = {2, 4, 6, 9, 12, 16, 18};
    for (int w : widths) {
        if (w >= g->tiles && (size_t)2 * 3 * w * 1024 + (size_t)g->chunks * 64 + (size_t)w * 64 + 256 <= LDS16_MAX) {
            g->parts = 1; g->tpp = w; g->stream = 1;
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

constexpr int PACK16_MAX_JOBS = 24;      // 24 x ~100 bytes of kernel arguments (limit 4 KB)
struct Pack16Jobs {
    Pack16Args job[PACK16_MAX_JOBS];
    int first_block[PACK16_MAX_JOBS + 1];    // job q owns blocks [first_block[q], first_block[q+1])
    int njobs;
};

// one thread per (part, chunk, row, k) element of an image; writes the three pieces.  Up to PACK16_MAX_JOBS images per launch.
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
    int paced;                 // two-part launches: twin waves of the two parts start every row tile together
    int pace_slot;             // which copy of the progress counters a paced launch uses
    const int* gate; int gate_run_if;      // optional device word: the launch does its work only when (*gate != 0) == gate_run_if
};
#define HGV_GATE_EXIT16(a) if ((a).gate != nullptr && ((*(a).gate != 0) ? 1 : 0) != (a).gate_run_if) return

// Progress counters of paced two-part launches: [part][workgroup][wave] = row tiles whose loads have begun.  They are
// hints only (a stale or missing value costs time, never correctness), so launches may share them.
constexpr int PACE_WGS = 256, PACE_WAVES = 16, PACE_SPINS = 512, PACE_AHEAD = 2;
constexpr int64_t PACE_MIN_BYTES = (int64_t)256 << 20;
__device__ int pace_ctr[HGV_PACE_SLOTS * 2 * PACE_WGS * PACE_WAVES];

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

    // Paced launches: workgroup r of part 0 and workgroup r of part 1 sit on the same XCD and their waves walk the
    // same row tiles.  Every wave publishes how many load groups it has issued and lets its twin be at most PACE_AHEAD
    // groups behind, so the second of the two reads of every A line finds it in the XCD's L2 instead of fetching it from
    // memory again.  The twin's counter is read two steps before it is looked at (by then the wave waits for older
    // loads anyway), so the wave that is behind never waits for it; the one that is ahead polls.  A twin that does
    // not catch up within PACE_SPINS polls (not resident, other XCD) switches this wave's pacing off for the launch.
    int* pace_base = pace_ctr + a.pace_slot * (2 * PACE_WGS * PACE_WAVES);
    int* pace_mine = pace_base + ((part * PACE_WGS + wg_rank) * PACE_WAVES + wib);
    const int* pace_twin = pace_base + (((part ^ 1) * PACE_WGS + wg_rank) * PACE_WAVES + wib);
    bool pace = a.paced != 0;
    int pace_step = 0;                                // load groups issued by this wave
    int pace_seen[RING];                              // the twin's counter as read by the last RING load groups (slot = ring slot)
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
        if (pace) pace_gate(d);
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
    // done: a twin that issues a few more load groups (other ring depth) must not wait for this wave
    if (a.paced && lane == 0) __hip_atomic_store(pace_mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One part per launch (the whole weight, or one part of a weight split along N).
template <int NT>
__global__ __launch_bounds__(64 * nt16_waves<NT>()) void k_gemm_nt16(const Nt16Args a, const int part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT16(a);
    nt16_body<NT, nt16_ring<NT>()>(a, lds_raw, part, blockIdx.x, gridDim.x);
}

// Both parts of a weight split along N in one launch: the first a.wgs0 workgroups keep part 0
// (NT0 n tiles) resident, the others part 1 (NT1 = NT0 or NT0 - 1 tiles); each group streams all rows of A.
template <int NT0, int NT1>
__global__ __launch_bounds__(64 * nt16_waves<NT0>()) void k_gemm_nt16_pair(const Nt16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT16(a);
    if ((int)blockIdx.x < a.wgs0) nt16_body<NT0, nt16_ring<NT0>()>(a, lds_raw, 0, blockIdx.x, a.wgs0);
    else nt16_body<NT1, nt16_ring<NT1>()>(a, lds_raw, 1, (int)blockIdx.x - a.wgs0, (int)gridDim.x - a.wgs0);
}

// ------------------------------------------------------------------------------------------
// Same product for weights whose split image does not fit in LDS (hidden width 256): the image is
// streamed.  A workgroup of 12 waves owns blocks of 12 row tiles (one per wave, all NT n tiles of it);
// for every 32-wide k chunk the waves multiply from one LDS buffer while all threads copy the next
// chunk of the image (3 pieces x 16 NT rows x 64 B, contiguous in the packed image and L2-resident)
// into the other; one barrier per chunk.  A is loaded and split per wave as in k_gemm_nt16, one chunk
// ahead.
template <int NT>
__global__ __launch_bounds__(768) void k_gemm_nt16s(const Nt16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HGV_GATE_EXIT16(a);
    const Geo16& g = a.g;
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int ROWS = NT * 16, CHUNK_BYTES = 3 * ROWS * 64, UNITS = CHUNK_BYTES / 16, UPT = (UNITS + 767) / 768;
    int4* tab = reinterpret_cast<int4*>(lds_raw + 2 * CHUNK_BYTES);
    float* bias_s = reinterpret_cast<float*>(lds_raw + 2 * CHUNK_BYTES + (size_t)g.chunks * 64);
    for (int e = t; e < g.chunks * 4; e += 768) {
        int second = 0, col = 0; int64_t ld = a.lda1;
        if (e < g.nb1) { col = 8 * e; }
        else if (e < g.nb1 + g.nb2) { second = 1; col = 8 * (e - g.nb1); ld = a.lda2; }
        const int c0 = (int)min((int64_t)col, ld - 4), c1 = (int)min((int64_t)col + 4, ld - 4);
        tab[e] = make_int4(c0 * 4, c1 * 4, second, 0);
    }
    for (int e = t; e < ROWS; e += 768) bias_s[e] = (a.bias && e < g.N) ? a.bias[e] : 0.f;

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
    __syncthreads();                       // table, bias and the first chunk are in place

    // A of (row block, chunk): two 16-byte halves of the lane's 8 consecutive k
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
        for (int n = 0; n < NT; ++n) acc[n] = *reinterpret_cast<const f32x4*>(&bias_s[16 * n + 4 * kq]);
        for (int c = 0; c < chunks; ++c) {
            const int cn = c + 1 < chunks ? c + 1 : 0;
            w_load(cn);                                      // next chunk of the image (cyclic): global -> registers
            bf16x8 ah, am, al;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 h, m, l;
                split3(a_nx[j >> 2][j & 3], h, m, l);
                ah[j] = h; am[j] = m; al[j] = l;
            }
            // A of the next step: next chunk of this block, or chunk 0 of the workgroup's next block (clamped at the end)
            {
                const int64_t rbn = c + 1 < chunks ? rb : (rb + gridDim.x < nblocks ? rb + gridDim.x : rb);
                a_load(rbn, cn);
            }
            const unsigned char* base = lds_raw + buf * CHUNK_BYTES + wlane;
            constexpr int GS = 3;
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += GS) {
                bf16x8 wh[GS], wm[GS], wl[GS];
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    if (n0 + j < NT) {
                        const unsigned char* w = base + (size_t)(n0 + j) * 16 * 64;
                        wh[j] = *reinterpret_cast<const bf16x8*>(w);
                        wm[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)ROWS * 64);
                        wl[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)2 * ROWS * 64);
                    }
                }
#define HGV_S_TERM(WP, AP)                                                                                         \
                _Pragma("unroll") for (int j = 0; j < GS; ++j)                                                    \
                    if (n0 + j < NT) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WP[j], AP, acc[n0 + j], 0, 0, 0);
                HGV_S_TERM(wl, ah) HGV_S_TERM(wh, al) HGV_S_TERM(wm, am) HGV_S_TERM(wm, ah) HGV_S_TERM(wh, am) HGV_S_TERM(wh, ah)
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
                if (col < a.ldc) *reinterpret_cast<f32x4*>(a.C + row * a.ldc + col) = acc[n];
            }
        }
    }
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

// Fewest waves per workgroup: every workgroup first copies its weight image (up to 160 KB) into LDS, which two waves do
// in eight rounds of loads and four waves in four; with a few hundred row tiles in all (a 4 416-row batch) that copy is
// most of the launch (4 waves: 3.5 % of the captured chignolin step against 1-2; 8 the same, 12 slower).
constexpr int nt_min_waves() { return 4; }

template <int NT>
int launch_nt16(const Nt16Args& a, int part, size_t lds, hipStream_t st) {
    auto kern = k_gemm_nt16<NT>;
    const int rc = raise_lds16(kern);
    if (rc != HGV_OK) return rc;
    // one workgroup per compute unit; as many waves (<= 16) as there are tiles to hand out
    const int cus = cu_count16();
    int64_t wpb = hgv_cdiv(a.ntiles, cus);
    if (wpb < nt_min_waves()) wpb = nt_min_waves();
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

template <int NT>
int launch_nt16s_nt(const Nt16Args& a, hipStream_t st) {
    auto kern = k_gemm_nt16s<NT>;
    const int rc = raise_lds16(kern);
    if (rc != HGV_OK) return rc;
    const size_t lds = (size_t)2 * 3 * NT * 1024 + (size_t)a.g.chunks * 64 + (size_t)NT * 64;
    int64_t blocks = hgv_cdiv(a.ntiles, 12);
    if (blocks > cu_count16()) blocks = cu_count16();
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(768), lds, st, a);
    return HGV_OK;
}

int launch_nt16s(const Nt16Args& a, hipStream_t st) {
    switch (a.g.tpp) {
        case 2: return launch_nt16s_nt<2>(a, st);
        case 4: return launch_nt16s_nt<4>(a, st);
        case 6: return launch_nt16s_nt<6>(a, st);
        case 9: return launch_nt16s_nt<9>(a, st);
        case 12: return launch_nt16s_nt<12>(a, st);
        case 16: return launch_nt16s_nt<16>(a, st);
        case 18: return launch_nt16s_nt<18>(a, st);
        default: return HGV_EUNSUPPORTED;
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
    if (wpb < nt_min_waves()) wpb = nt_min_waves();
    if (wpb > nt16_waves<NT0>()) wpb = nt16_waves<NT0>();
    if (wpb < 1) wpb = 1;
    // Share of the compute units per part: a wave's time is (rounds of row tiles) x (cost of one row
    // tile), and the cost grows slowly with the part's width (measured: 33 / 36 / 44 us for 4 / 5 / 9
    // n tiles, i.e. ~ 11 + NT), so the split that minimises the slower part is near even and is
    // decided by the rounding of the rounds.  Ties go to the split closest to even.
    int share0 = cus / 2;
    // Rows that no longer stay in the memory-side cache between the two parts' reads: pace the parts so that the
    // second read hits L2 (needs the same walk over the row tiles in both parts, i.e. an even split, and twins on
    // one XCD: workgroups r and r + wgs0 with wgs0 a multiple of 8).
    const int64_t a_bytes = a.M * 4 * (a.lda1 + (a.A2 ? a.lda2 : 0));
    const bool paced = a_bytes >= PACE_MIN_BYTES && cus % 16 == 0 && cus / 2 <= PACE_WGS && wpb <= PACE_WAVES &&
                       hgv_cdiv(a.ntiles, wpb) >= cus / 2 && a.ntiles < ((int64_t)1 << 31);
    a.paced = paced ? 1 : 0;
    if (!paced) {
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

namespace {

// ------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores:  GW[n][k] = sum_m G[m][n] A[m][k]
// ------------------------------------------------------------------------------------------
// Both operands are reduced over their ROW index, so an MFMA lane needs 8 consecutive rows of one
// column, while coalesced loads deliver consecutive columns of one row.  A workgroup therefore stages
// 32-row tiles (144 columns of G, 144 columns of A) in LDS row-major as three bf16 pieces and reads
// the operands back with the transposing ds_read_b64_tr_b16.  LDS image of a piece: 18 panels of 16
// columns, a panel = 32 row slots of 32 bytes (+32 bytes of padding so that the four panels a 16-lane
// group writes fall into different banks); row r sits in slot r with bits 2 and 3 swapped, which makes
// the two 4-row blocks a 32-lane half reads (rows 8kq.. and 8(kq+1)..) cover all 64 banks once.
// Two buffers, one barrier per step: while the matrix cores work on one buffer every wave splits its
// share (3 float4 per thread) of the next step into the other.  Wave (wn = id % 3, wk = id / 3) owns
// n tiles 3wn..3wn+2 and k tiles {0,1,2}, {3,4}, {5,6}, {7,8} for wk = 0..3, which gives the four
// SIMDs (waves id, id+4, id+8 share one) 21, 21, 21 and 18 tiles; waves with wk == 1 also form the
// column sums of G (products with a tile of ones).  A workgroup covers one (n group, k group) of 144 x
// 144 outputs over a slice of the rows and writes its 81 accumulator tiles as a partial set;
// k_reduce_tn16 adds the sets in fp64 in a fixed order and writes (or accumulates into) the final
// matrices.
#ifndef T16_INTERLEAVE
#define T16_INTERLEAVE 1
#endif
constexpr int T16_ROWS = 32;
constexpr int T16_PANEL = 1024 + 32;
constexpr int T16_PIECE = 18 * T16_PANEL;
constexpr int T16_BUF = 3 * T16_PIECE;
constexpr int T16_THREADS = 768;
constexpr int T16_SET_FLOATS = 81 * 256;

// Progress counters of paced weight-gradient launches, one per workgroup (64 bytes apart): 32-row steps finished.
// Hints only, like pace_ctr above.
constexpr int PACE_TN_AHEAD = 1;
__device__ int pace_tn[HGV_PACE_SLOTS * PACE_WGS * 16];

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Tn16Args {
    const float* G; int64_t ldg; int N;
    const float* A[2]; int64_t lda[2]; int K[2];
    int kg0;                   // k groups (144 columns each) of the first source
    int ngrps, kgrps;          // groups over N and over both sources' K
    int64_t M;
    int64_t steps;             // 32-row steps in total = ceil(M / 32), M >= 32
    int sets;                  // row slices per group pair
    float* partial;            // [ngrp][kgrp][set][81][4][64]
    float* bias_partial;       // [ngrp][set][9][4][64], written by the workgroups of k group 0 (null: no bias)
    int paced;                 // two k groups, one n group: the two workgroups of a row slice share an XCD and walk G in step
    int pace_slot;             // which copy of the progress counters a paced launch uses
    const float* G2; int64_t ldg2; int n_split;   // optional: columns >= n_split (a multiple of 4) of G come from G2 (column n - n_split)
};

__device__ __forceinline__ int t16_slot(int r) { return (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ bf16x8 t16_read(const unsigned char* lds, int addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr + 256));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int KT>
__device__ __forceinline__ void tn16_body(const Tn16Args& a, unsigned char* lds, const int wn, const int k0t, const bool bias, const bool lead) {
    const int t = threadIdx.x, lane = t & 63;
    const int per = a.ngrps * a.kgrps;
    // paced: blocks b and b + 8 (same XCD under the round-robin dispatch) are the two k groups of one row slice
    const int grp = a.paced ? (int)(blockIdx.x >> 3) & 1 : (int)(blockIdx.x % per);
    const int set = a.paced ? (int)(blockIdx.x >> 4) * 8 + (int)(blockIdx.x & 7) : (int)(blockIdx.x / per);
    const int ngrp = grp / a.kgrps, kgrp = grp - ngrp * a.kgrps;
    const int second = kgrp >= a.kg0;
    const int n_off = 144 * ngrp, k_off = 144 * (second ? kgrp - a.kg0 : kgrp);
    const float* Ab = a.A[second]; const int64_t lda = a.lda[second];
    const int n_w = min(144, a.N - n_off), k_w = min(144, a.K[second] - k_off);     // live widths of the two tiles
    const int64_t s_lo = a.steps * set / a.sets, s_hi = a.steps * (set + 1) / a.sets;
    // staging: float4 f = t + 768 u of the step's 64 x 36 (32 rows of G, then 32 rows of A).  Columns past
    // the live width (rounded up to 4) re-read column 0: they only reach outputs that are never stored.
    const float* src[3]; int64_t ld[3]; int waddr[3]; int srow[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int f = t + T16_THREADS * u;
        const int isa = f >= 1152, rem = isa ? f - 1152 : f;
        const int row = rem / 36, c4 = rem - 36 * row, col = 4 * c4;
        const int width = isa ? k_w : n_w;
        const int ceff = col < (width + 3) / 4 * 4 ? col : 0;
        const bool g2 = !isa && a.G2 != nullptr && n_off + ceff >= a.n_split;
        ld[u] = isa ? lda : (g2 ? a.ldg2 : a.ldg);
        srow[u] = row;
        src[u] = (isa ? Ab + k_off : (g2 ? a.G2 - a.n_split : a.G) + n_off) + (int64_t)row * ld[u] + ceff;
        waddr[u] = (isa * 9 + (col >> 4)) * T16_PANEL + t16_slot(row) * 32 + (col & 15) * 2;
    }
    f32x4 stage[3];
    auto load_step = [&](int64_t s) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) stage[u] = *reinterpret_cast<const f32x4*>(src[u] + min(s * T16_ROWS, a.M - T16_ROWS) * ld[u]);
    };
    // The last step of a matrix whose row count is not a multiple of 32 re-reads the last 32 rows (load_step);
    // the first `skip` of them belong to the step before and are staged as zeros.
    auto write_step = [&](int b, int64_t s) __attribute__((always_inline)) {
        unsigned char* base = lds + b * T16_BUF;
        const int skip = (int)(s * T16_ROWS - min(s * T16_ROWS, a.M - T16_ROWS));        // scalar; 0 for whole steps
        if (skip > 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (srow[u] < skip) stage[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 hh, mm, ll; split3(stage[u][e], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
            *reinterpret_cast<bf16x4*>(base + waddr[u]) = h;
            *reinterpret_cast<bf16x4*>(base + T16_PIECE + waddr[u]) = m;
            *reinterpret_cast<bf16x4*>(base + 2 * T16_PIECE + waddr[u]) = l;
        }
    };
    // operand read address inside a panel: lane 4q+p of the 16-lane group kq supplies row 8kq+q, columns 4p..4p+3
    const int kq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int raddr = t16_slot(8 * kq + q) * 32 + p * 8;

    f32x4 acc[3][KT], accb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    if (s_lo < s_hi) {
        load_step(s_lo);
        write_step(0, s_lo);
        if (s_lo + 1 < s_hi) load_step(s_lo + 1);
    }
    __syncthreads();
    // Paced launches: the two workgroups of a row slice read the same rows of G.  Wave 0 publishes the steps its workgroup
    // has finished and holds the workgroup back at the step barrier while the twin is more than PACE_TN_AHEAD steps
    // behind, so that the later of the two reads of a G line finds it in the XCD's L2.  The twin's counter is loaded at
    // the top of a step and looked at after the step's MFMAs; a twin that does not catch up within PACE_SPINS polls
    // switches the pacing off for the rest of the launch.
    int* pace_mine = pace_tn + (a.pace_slot * PACE_WGS + blockIdx.x) * 16;
    const int* pace_twin = pace_tn + (a.pace_slot * PACE_WGS + (blockIdx.x ^ 8)) * 16;
    bool pace = a.paced != 0 && lead;
    int64_t s = s_lo;
#if T16_INTERLEAVE
    // Steady steps (s + 2 exists, so neither the staged nor the loaded step is the ragged last one): the split of the next
    // step's 12 values is cut into 36 slices of 3-4 VALU instructions, one behind each of the first 36 products, with the
    // order fixed - a matrix-core instruction occupies its pipe for 16 cycles and the slice issues meanwhile.  Left to itself
    // the compiler puts the whole split in front of an unbroken run of 36-54 products, every wave of the workgroup is in the
    // same phase (one barrier per step), and the step costs the sum of the two.
    for (; s + 2 < s_hi; ++s) {
        const int b = (int)((s - s_lo) & 1);
        const unsigned char* base = lds + b * T16_BUF;
        unsigned char* wbase = lds + (b ^ 1) * T16_BUF;
        int pace_seen = 0;
        if (pace) pace_seen = __hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bf16x8 g[3][3], x[KT][3];      // [tile][piece]
        // in the order the products need them (LDS returns in order: the first term starts after a third of the reads; the
        // reads of a step are ~2 000 cycles of the compute unit's LDS pipe, as much as the products are of a SIMD's matrix pipe)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int gp = o == 0 ? 2 : o - 1, xp = o == 0 ? 0 : 3 - o;          // (g2, x0), (g0, x2), (g1, x1)
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i][gp] = t16_read(base, gp * T16_PIECE + (3 * wn + i) * T16_PANEL + raddr);
#pragma unroll
            for (int j = 0; j < KT; ++j) x[j][xp] = t16_read(base, xp * T16_PIECE + (9 + k0t + j) * T16_PANEL + raddr);
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int GP[6] = {2, 0, 1, 0, 1, 0}, XP[6] = {0, 2, 1, 1, 0, 0};      // small terms first
        bf16x4 h4, m4, l4;
        float r1 = 0.f;
#pragma unroll
        for (int idx = 0; idx < 6 * 3 * KT; ++idx) {
            const int term = idx / (3 * KT), ij = idx - term * 3 * KT, i = ij / KT, j = ij - i * KT;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][GP[term]], x[j][XP[term]], acc[i][j], 0, 0, 0);
            if (idx < 36) {
                const int v = idx / 3, u = v >> 2, e = v & 3, part = idx - 3 * v;
                if (part == 0) { h4[e] = (__bf16)stage[u][e]; r1 = stage[u][e] - (float)h4[e]; }
                else if (part == 1) { m4[e] = (__bf16)r1; r1 = r1 - (float)m4[e]; }
                else {
                    l4[e] = (__bf16)r1;
                    if (e == 3) {
                        *reinterpret_cast<bf16x4*>(wbase + waddr[u]) = h4;
                        *reinterpret_cast<bf16x4*>(wbase + T16_PIECE + waddr[u]) = m4;
                        *reinterpret_cast<bf16x4*>(wbase + 2 * T16_PIECE + waddr[u]) = l4;
                        // the float4 is spent: its successor (step s + 2) leaves now, a step ahead of its own split (at the end
                        // of the step it would be wanted again a few hundred cycles later: 31 % of the wave cycles waited for it)
                        stage[u] = *reinterpret_cast<const f32x4*>(src[u] + min((s + 2) * T16_ROWS, a.M - T16_ROWS) * ld[u]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pc = 2; pc >= 0; --pc) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][pc], ones, accb[i], 0, 0, 0);
        }
        if (pace) {
            const int done = (int)(s - s_lo) + 1;
            if (lane == 0) __hip_atomic_store(pace_mine, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(pace_seen) < done - PACE_TN_AHEAD) {
                for (int spin = 0;; ++spin) {
                    const int seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (seen >= done - PACE_TN_AHEAD) break;
                    if (spin >= PACE_SPINS) { pace = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
    }
#endif
    for (; s < s_hi; ++s) {
        const int b = (int)((s - s_lo) & 1);
        const unsigned char* base = lds + b * T16_BUF;
        int pace_seen = 0;
        if (pace) pace_seen = __hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bf16x8 g[3][3], x[KT][3];      // [tile][piece]
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i][pc] = t16_read(base, pc * T16_PIECE + (3 * wn + i) * T16_PANEL + raddr);
#pragma unroll
            for (int j = 0; j < KT; ++j) x[j][pc] = t16_read(base, pc * T16_PIECE + (9 + k0t + j) * T16_PANEL + raddr);
        }
        // next step's tiles: split and store into the other buffer while the matrix cores work
        if (s + 1 < s_hi) write_step(b ^ 1, s + 1);
        if (s + 2 < s_hi) load_step(s + 2);
        // six piece products, small terms first: gl*xh, gh*xl, gm*xm, gh*xm, gm*xh, gh*xh
#define HGV_TN_TERM(GP, XP)                                                                             \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < KT; ++j)                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][GP], x[j][XP], acc[i][j], 0, 0, 0);
        HGV_TN_TERM(2, 0) HGV_TN_TERM(0, 2) HGV_TN_TERM(1, 1) HGV_TN_TERM(0, 1) HGV_TN_TERM(1, 0) HGV_TN_TERM(0, 0)
#undef HGV_TN_TERM
        if (bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pc = 2; pc >= 0; --pc) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[i][pc], ones, accb[i], 0, 0, 0);
        }
        if (pace) {
            const int done = (int)(s - s_lo) + 1;
            if (lane == 0) __hip_atomic_store(pace_mine, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(pace_seen) < done - PACE_TN_AHEAD) {
                for (int spin = 0;; ++spin) {
                    const int seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (seen >= done - PACE_TN_AHEAD) break;
                    if (spin >= PACE_SPINS) { pace = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
    }
    if (a.paced != 0 && lead && lane == 0) __hip_atomic_store(pace_mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // D layout: lane (j = lane & 15, q = lane >> 4), register r: n = 16 tn + 4 q + r, k = 16 tk + j
    float* out = a.partial + ((int64_t)grp * a.sets + set) * T16_SET_FLOATS;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(((3 * wn + i) * 9 + k0t + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    if (bias && a.bias_partial && kgrp == 0) {
        float* bo = a.bias_partial + ((int64_t)ngrp * a.sets + set) * (9 * 256);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) bo[((3 * wn + i) * 4 + r) * 64 + lane] = accb[i][r];
    }
}

__global__ __launch_bounds__(T16_THREADS) void k_gemm_tn16(const Tn16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wk = w / 3, wn = w - 3 * wk;
    if (wk == 0) tn16_body<3>(a, lds_raw, wn, 0, false, w == 0);
    else tn16_body<2>(a, lds_raw, wn, 1 + 2 * wk, wk == 1, false);
}

// ------------------------------------------------------------------------------------------
// The same weight gradient from two scaled fp16 pieces per operand and three piece products ("f16x3").
// Both operands are reduced over their rows, so a scale has to be constant over the rows a workgroup adds up: each
// workgroup keeps ONE power-of-two scale for its tiles of G and one for its tiles of A, such that the largest entry seen
// so far sits in [2^12, 2^13).  Every wave publishes the largest |value| of the 32-row tile it has just loaded (LDS
// atomic max, order-independent) one step before that tile is split, so the scale is exact, not predicted; when a
// tile would leave the fp16 range with the running scale, the scale is lowered and the accumulators are multiplied by
// the ratio (a power of two).  Entries within 2^-16 of the largest keep all 22 bits, smaller ones an absolute accuracy
// of 2^-25 on the 2^12 scale.  Partial sets are written unscaled (multiplied by 1 / (s_g s_a), exact), so the
// reduction kernel is shared with the bf16 path.  Staging layout and wave roles are those of k_gemm_tn16.
typedef _Float16 f16x8t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4t __attribute__((ext_vector_type(4)));
constexpr int T16H_BUF = 2 * T16_PIECE;
constexpr int T16H_LDS = 2 * T16H_BUF + 64;

__device__ __forceinline__ float scale_for_t(float amax, float& inv) {
    const unsigned e = __float_as_uint(amax) >> 23;
    const unsigned se = min(266u - e, 254u);
    inv = __uint_as_float((254u - se) << 23);
    return __uint_as_float(se << 23);
}

// largest value of a wave (all lanes receive it); x >= 0
__device__ __forceinline__ float wave_max(float x) {
    int v = __float_as_int(x);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));     // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));     // row_mirror
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)r[0], (int)r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return __int_as_float(max((int)q[0], (int)q[1]));
}

__device__ __forceinline__ float amax4(const f32x4& v) {
    return fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
}

__device__ __forceinline__ f16x8t t16h_read(const unsigned char* lds, int addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + addr + 256));
    union { struct { s16x4 a, b; } s; f16x8t v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int KT>
__device__ __forceinline__ void tn16h_body(const Tn16Args& a, unsigned char* lds, const int wn, const int k0t, const bool bias, const bool lead) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int per = a.ngrps * a.kgrps;
    const int grp = a.paced ? (int)(blockIdx.x >> 3) & 1 : (int)(blockIdx.x % per);
    const int set = a.paced ? (int)(blockIdx.x >> 4) * 8 + (int)(blockIdx.x & 7) : (int)(blockIdx.x / per);
    const int ngrp = grp / a.kgrps, kgrp = grp - ngrp * a.kgrps;
    const int second = kgrp >= a.kg0;
    const int n_off = 144 * ngrp, k_off = 144 * (second ? kgrp - a.kg0 : kgrp);
    const float* Ab = a.A[second]; const int64_t lda = a.lda[second];
    const int n_w = min(144, a.N - n_off), k_w = min(144, a.K[second] - k_off);
    const int64_t s_lo = a.steps * set / a.sets, s_hi = a.steps * (set + 1) / a.sets;
    const float* src[3]; int64_t ld[3]; int waddr[3]; int srow[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int f = t + T16_THREADS * u;
        const int isa = f >= 1152, rem = isa ? f - 1152 : f;
        const int row = rem / 36, c4 = rem - 36 * row, col = 4 * c4;
        const int width = isa ? k_w : n_w;
        const int ceff = col < (width + 3) / 4 * 4 ? col : 0;
        ld[u] = isa ? lda : a.ldg;
        srow[u] = row;
        src[u] = (isa ? Ab + k_off : a.G + n_off) + (int64_t)row * ld[u] + ceff;
        waddr[u] = (isa * 9 + (col >> 4)) * T16_PANEL + t16_slot(row) * 32 + (col & 15) * 2;
    }
    // float4 u = 0 of every thread belongs to G, u = 2 to A, u = 1 to G in waves 0-5 and to A in waves 6-11
    const bool mid_is_g = wave < 6;
    unsigned* slot = reinterpret_cast<unsigned*>(lds + 2 * T16H_BUF);      // [3][2]: largest |G|, |A| of tile s in slot s % 3
    // Two sets of staging registers: the tile split in step s (tile s + 1) was loaded two steps earlier and its largest
    // value published one step earlier, so neither the publication nor the split ever waits for a load just issued.
    f32x4 stage[2][3];
    auto load_step = [&](int64_t s, int set_) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) stage[set_][u] = *reinterpret_cast<const f32x4*>(src[u] + min(s * T16_ROWS, a.M - T16_ROWS) * ld[u]);
    };
    auto publish = [&](int64_t s, int set_) __attribute__((always_inline)) {
        const float m0 = amax4(stage[set_][0]), m1 = amax4(stage[set_][1]), m2 = amax4(stage[set_][2]);
        const float mg = wave_max(mid_is_g ? fmaxf(m0, m1) : m0), ma = wave_max(mid_is_g ? m2 : fmaxf(m1, m2));
        if (lane == 0) {
            unsigned* sl = slot + 2 * (int)(s % 3);
            __hip_atomic_fetch_max(sl, __float_as_uint(mg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(sl + 1, __float_as_uint(ma), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    float sg = 1.f, sa = 1.f, ig = 1.f, ia = 1.f;        // running scales of G and A and their inverses (same in every thread)
    // scale for tile s (published before the last barrier): lowered when the tile would pass 2^15; returns acc multipliers
    auto consume = [&](float mg, float ma, bool first, float& rg, float& ra) __attribute__((always_inline)) {
        rg = 1.f; ra = 1.f;
        if (first || mg * sg > 32768.f) { float inv; const float sn = scale_for_t(mg, inv); rg = sn * ig; sg = sn; ig = inv; }
        if (first || ma * sa > 32768.f) { float inv; const float sn = scale_for_t(ma, inv); ra = sn * ia; sa = sn; ia = inv; }
    };
    auto write_step = [&](int b, int64_t s, int set_) __attribute__((always_inline)) {
        unsigned char* base = lds + b * T16H_BUF;
        const int skip = (int)(s * T16_ROWS - min(s * T16_ROWS, a.M - T16_ROWS));
        if (skip > 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (srow[u] < skip) stage[set_][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float sc = u == 0 ? sg : (u == 2 ? sa : (mid_is_g ? sg : sa));
            f16x4t h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = stage[set_][u][e] * sc;
                const _Float16 hh = (_Float16)y;
                h[e] = hh; l[e] = (_Float16)(y - (float)hh);
            }
            *reinterpret_cast<f16x4t*>(base + waddr[u]) = h;
            *reinterpret_cast<f16x4t*>(base + T16_PIECE + waddr[u]) = l;
        }
    };
    const int kq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int raddr = t16_slot(8 * kq + q) * 32 + p * 8;

    f32x4 acc[3][KT], accb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f16x8t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

    if (t < 6) slot[t] = 0u;
    __syncthreads();
    if (s_lo < s_hi) {
        load_step(s_lo, 0);
        publish(s_lo, 0);
    }
    __syncthreads();
    if (s_lo < s_hi) {
        float rg, ra;
        consume(__uint_as_float(slot[2 * (int)(s_lo % 3)]), __uint_as_float(slot[2 * (int)(s_lo % 3) + 1]), true, rg, ra);
        write_step(0, s_lo, 0);
        if (s_lo + 1 < s_hi) { load_step(s_lo + 1, 1); publish(s_lo + 1, 1); }
        if (s_lo + 2 < s_hi) load_step(s_lo + 2, 0);
    }
    __syncthreads();
    int* pace_mine = pace_tn + (a.pace_slot * PACE_WGS + blockIdx.x) * 16;
    const int* pace_twin = pace_tn + (a.pace_slot * PACE_WGS + (blockIdx.x ^ 8)) * 16;
    bool pace = a.paced != 0 && lead;
    // step s with PAR = (s - s_lo) & 1: tile s + 1 sits in register set PAR ^ 1, tile s + 2 in set PAR
    auto step = [&](int64_t s, const int PAR) __attribute__((always_inline)) {
        const int b = PAR;
        const unsigned char* base = lds + b * T16H_BUF;
        int pace_seen = 0;
        if (pace) pace_seen = __hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the next tile's largest values first: LDS answers in order, and the split below needs only these two
        const int sn3 = (int)((s + 1) % 3);
        const float mg_next = __uint_as_float(slot[2 * sn3]), ma_next = __uint_as_float(slot[2 * sn3 + 1]);
        f16x8t g[3][2], x[KT][2];      // [tile][piece]
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i][pc] = t16h_read(base, pc * T16_PIECE + (3 * wn + i) * T16_PANEL + raddr);
#pragma unroll
            for (int j = 0; j < KT; ++j) x[j][pc] = t16h_read(base, pc * T16_PIECE + (9 + k0t + j) * T16_PANEL + raddr);
        }
        // next step's tiles: scale (published a step ago), split and store into the other buffer while the matrix cores work
        float rg = 1.f, ra = 1.f;
        if (s + 1 < s_hi) {
            consume(mg_next, ma_next, false, rg, ra);
            write_step(b ^ 1, s + 1, PAR ^ 1);
        }
        if (t == 0) { slot[2 * (int)(s % 3)] = 0u; slot[2 * (int)(s % 3) + 1] = 0u; }      // read a step ago, published again two steps on
        load_step(min(s + 3, s_hi - 1), PAR ^ 1);          // unconditional (past the end: the last tile again), so that waits stay counted
        // largest values of the tile after next, in front of the products: the LDS atomics then have the whole product
        // phase to finish before the barrier
        if (s + 2 < s_hi) publish(s + 2, PAR);
        // three piece products, small terms first: gl*xh, gh*xl, gh*xh
#define HGV_TNH_TERM(GP, XP)                                                                            \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < KT; ++j)                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g[i][GP], x[j][XP], acc[i][j], 0, 0, 0);
        HGV_TNH_TERM(1, 0) HGV_TNH_TERM(0, 1) HGV_TNH_TERM(0, 0)
#undef HGV_TNH_TERM
        if (bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pc = 1; pc >= 0; --pc) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g[i][pc], ones, accb[i], 0, 0, 0);
        }
        if (rg != 1.f || ra != 1.f) {              // uniform: the next tile is accumulated on a lower scale
            const float r2 = rg * ra;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < KT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] *= r2;
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[i][r] *= rg;
            }
        }
        if (pace) {
            const int done = (int)(s - s_lo) + 1;
            if (lane == 0) __hip_atomic_store(pace_mine, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(pace_seen) < done - PACE_TN_AHEAD) {
                for (int spin = 0;; ++spin) {
                    const int seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pace_twin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (seen >= done - PACE_TN_AHEAD) break;
                    if (spin >= PACE_SPINS) { pace = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
    };
    for (int64_t s = s_lo; s < s_hi; s += 2) {
        step(s, 0);
        if (s + 1 < s_hi) step(s + 1, 1);
    }
    if (a.paced != 0 && lead && lane == 0) __hip_atomic_store(pace_mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const float unscale = ig * ia;
    float* out = a.partial + ((int64_t)grp * a.sets + set) * T16_SET_FLOATS;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(((3 * wn + i) * 9 + k0t + j) * 4 + r) * 64 + lane] = acc[i][j][r] * unscale;
    if (bias && a.bias_partial && kgrp == 0) {
        float* bo = a.bias_partial + ((int64_t)ngrp * a.sets + set) * (9 * 256);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) bo[((3 * wn + i) * 4 + r) * 64 + lane] = accb[i][r] * ig;
    }
}

__global__ __launch_bounds__(T16_THREADS) void k_gemm_tn16h(const Tn16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wk = w / 3, wn = w - 3 * wk;
    if (wk == 0) tn16h_body<3>(a, lds_raw, wn, 0, false, w == 0);
    else tn16h_body<2>(a, lds_raw, wn, 1 + 2 * wk, wk == 1, false);
}

struct Tr16Args {
    Tn16Args t;
    float* GW[2]; int64_t ldgw[2];
    float* gbias;
    int accumulate;
};

// One workgroup: 64 consecutive floats of a partial set (coalesced) x 4 interleaved groups of sets.
__global__ __launch_bounds__(256) void k_reduce_tn16(const Tr16Args r) {
    __shared__ double sh[4][65];
    const Tn16Args& a = r.t;
    const int el = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t tile_total = (int64_t)a.ngrps * a.kgrps * T16_SET_FLOATS;
    const int64_t e = (int64_t)blockIdx.x * 64 + el;
    const bool is_bias = e >= tile_total;
    const int64_t eb = e - tile_total;
    const bool live = !is_bias || (r.gbias != nullptr && eb < (int64_t)a.ngrps * 9 * 256);
    double s = 0.0;
    int grp = 0, off = 0;
    if (live) {
        const float* src;
        int64_t stride;
        if (!is_bias) {
            grp = (int)(e / T16_SET_FLOATS); off = (int)(e - (int64_t)grp * T16_SET_FLOATS);
            src = a.partial + (int64_t)grp * a.sets * T16_SET_FLOATS + off; stride = T16_SET_FLOATS;
        } else {
            grp = (int)(eb / (9 * 256)); off = (int)(eb - (int64_t)grp * 9 * 256);
            src = a.bias_partial + (int64_t)grp * a.sets * (9 * 256) + off; stride = 9 * 256;
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = part;
        for (; b + 28 < a.sets; b += 32) {                 // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 4 * u) * stride];
            s0 += (double)v[0] + (double)v[4]; s1 += (double)v[1] + (double)v[5];
            s2 += (double)v[2] + (double)v[6]; s3 += (double)v[3] + (double)v[7];
        }
        for (; b < a.sets; b += 4) s0 += (double)src[(int64_t)b * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[part][el] = s;
    __syncthreads();
    if (part != 0 || !live) return;
    double v = ((sh[0][el] + sh[1][el]) + sh[2][el]) + sh[3][el];
    const int lane = off & 63, rr = (off >> 6) & 3, tile = off >> 8;
    if (is_bias) {
        if ((lane & 15) != 0) return;                            // every column of a ones product holds the same sum
        const int n = 144 * grp + 16 * tile + 4 * (lane >> 4) + rr;
        if (tile >= 9 || n >= a.N || n >= 144 * (grp + 1)) return;
        r.gbias[n] = r.accumulate ? r.gbias[n] + (float)v : (float)v;
    } else {
        const int ngrp = grp / a.kgrps, kgrp = grp - ngrp * a.kgrps;
        const int second = kgrp >= a.kg0;
        const int tn = tile / 9, tk = tile - 9 * tn;
        const int nl = 16 * tn + 4 * (lane >> 4) + rr, kl = 16 * tk + (lane & 15);
        const int n = 144 * ngrp + nl, k = 144 * (second ? kgrp - a.kg0 : kgrp) + kl;
        if (n >= a.N || k >= a.K[second]) return;
        float* dst = r.GW[second] + (int64_t)n * r.ldgw[second] + k;
        *dst = r.accumulate ? *dst + (float)v : (float)v;
    }
}

struct Tn16Plan { int ngrps, kg0, kgrps, sets; int64_t steps; bool ok; };

// G bytes from which the weight-gradient launch is paced (HGVAE_TN_PACE_MIN_MB: lab override)
int64_t tn_pace_min_bytes() {
    static const int64_t v = [] {
        const char* e = getenv("HGVAE_TN_PACE_MIN_MB");
        return e ? (int64_t)atoll(e) << 20 : PACE_MIN_BYTES;
    }();
    return v;
}

// Shapes worth the staging: rows for at least a few steps per workgroup and tiles that are mostly live.
Tn16Plan tn16_plan(int64_t M, int N, int K1, int K2) {
    Tn16Plan p{};
    p.ngrps = (int)hgv_cdiv(N, 144);
    p.kg0 = (int)hgv_cdiv(K1, 144);
    p.kgrps = p.kg0 + (K2 > 0 ? (int)hgv_cdiv(K2, 144) : 0);
    p.steps = hgv_cdiv(M, T16_ROWS);
    const int per = p.ngrps * p.kgrps;
    int64_t sets = cu_count16() / per;
    if (sets < 1) sets = 1;
    if (sets > p.steps / 2) sets = p.steps / 2;        // at least two 32-row steps per workgroup (a few thousand rows: 2 % of a captured step against four)
    p.sets = (int)sets;
    const double live = (double)N * (K1 + K2) / ((double)per * 144 * 144);
    p.ok = N > 0 && K1 > 0 && K2 >= 0 && p.sets >= 1 && live >= 0.6 && p.ngrps <= 2;
    return p;
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

// Up to 24 images in one launch: job q packs [W1[q] | W2[q]] (W2[q] may be null, then K2[q] = 0) into out[q].
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
static int gemm_nt16_any(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                         const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int gate_run_if,
                         hipStream_t stream) {
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
    a.gate = gate; a.gate_run_if = gate_run_if;
    a.pace_slot = hgv_pace_slot(stream);
    const size_t lds = part_bytes16(a.g.tpp, a.g.chunks) + (size_t)a.g.chunks * 64 + 9 * 16 * 4;
    // a part holds exactly its n tiles; the rows of A are streamed once per part
    int rc;
    if (a.g.stream) {
        rc = launch_nt16s(a, stream);
    } else if (a.g.parts == 1) {
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

extern "C" int hgv_gemm_nt16(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                             const float* bias, int64_t M, int N, float* C, int64_t ldc, hipStream_t stream) {
    return gemm_nt16_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, nullptr, 0, stream);
}

// hgv_gemm_nt16 that does its work only when (*gate != 0) == run_if_set (gate: device int written earlier on the
// stream); otherwise every workgroup returns at once and C is left untouched.
extern "C" int hgv_gemm_nt16_gated(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                                   const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int run_if_set,
                                   hipStream_t stream) {
    HGV_REQUIRE(gate != nullptr, HGV_EINVAL, "gemm_nt16_gated: null gate");
    return gemm_nt16_any(A1, lda1, K1, A2, lda2, K2, Wp, bias, M, N, C, ldc, gate, run_if_set ? 1 : 0, stream);
}

// ---- weight gradients on the bf16 matrix cores (same contract as hgv_gemm_tn / hgv_gemm_tn2) --------------------
// 1 when hgv_gemm_tn16 runs this shape (enough rows, tiles mostly live); buffers must be 16-byte aligned with
// leading dimensions that are multiples of 4.
extern "C" int hgv_gemm_tn16_supported(int64_t M, int N, int K1, int K2) { return tn16_plan(M, N, K1, K2).ok ? 1 : 0; }

extern "C" size_t hgv_gemm_tn16_ws(int64_t M, int N, int K1, int K2) {
    const Tn16Plan p = tn16_plan(M, N, K1, K2);
    if (!p.ok) return 256;
    return hgv_align256((size_t)p.ngrps * p.kgrps * p.sets * T16_SET_FLOATS * 4) + hgv_align256((size_t)p.ngrps * p.sets * 9 * 256 * 4);
}

// GW1[N,K1] (+)= G^T A1, GW2[N,K2] (+)= G^T A2 (A2 / GW2 null for a single product), gbias[N] (+)= column sums of G.
static int gemm_tn16_any(bool f16, const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                        const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias,
                        int accumulate, void* ws, size_t ws_bytes, hipStream_t stream, const float* Gb = nullptr,
                        int64_t ldgb = 0, int n_split = 0) {
    if (!A2) K2 = 0;
    HGV_REQUIRE(M > 0 && N > 0 && K1 > 0 && (A2 ? K2 > 0 && GW2 : true), HGV_EINVAL, "gemm_tn16: bad sizes");
    const Tn16Plan p = tn16_plan(M, N, K1, K2);
    HGV_REQUIRE(p.ok, HGV_EUNSUPPORTED, "gemm_tn16: shape %lld x %d x (%d + %d) is not supported", (long long)M, N, K1, K2);
    HGV_REQUIRE(G && A1 && GW1 && ws && hgv_aligned(G, 16) && hgv_aligned(A1, 16) && ldg % 4 == 0 && lda1 % 4 == 0 &&
                ldg >= (Gb ? n_split : N) && lda1 >= K1 && ldgw1 >= K1, HGV_EINVAL, "gemm_tn16: null / misaligned buffer or bad leading dimension");
    if (A2) HGV_REQUIRE(hgv_aligned(A2, 16) && lda2 % 4 == 0 && lda2 >= K2 && ldgw2 >= K2, HGV_EINVAL,
                        "gemm_tn16: misaligned second source or bad leading dimension");
    HGV_REQUIRE(ws_bytes >= hgv_gemm_tn16_ws(M, N, K1, K2) && hgv_aligned(ws, 256), HGV_EWORKSPACE, "gemm_tn16: workspace too small or misaligned");
    Tr16Args r{};
    Tn16Args& a = r.t;
    a.G = G; a.ldg = ldg; a.N = N;
    a.G2 = Gb; a.ldg2 = ldgb; a.n_split = n_split;
    if (Gb) HGV_REQUIRE(!f16 && n_split > 0 && n_split % 4 == 0 && n_split < N && hgv_aligned(Gb, 16) && ldgb % 4 == 0 && ldgb >= N - n_split,
                        HGV_EINVAL, "gemm_tn16_gg: bad second gradient matrix");
    a.A[0] = A1; a.lda[0] = lda1; a.K[0] = K1; a.A[1] = A2 ? A2 : A1; a.lda[1] = A2 ? lda2 : lda1; a.K[1] = K2;
    a.kg0 = p.kg0; a.ngrps = p.ngrps; a.kgrps = p.kgrps; a.M = M; a.steps = p.steps; a.sets = p.sets;
    // pacing pays once G no longer stays in the memory-side cache between its two reads; it needs every workgroup resident
    // (one per CU) and whole groups of 16 blocks
    a.pace_slot = hgv_pace_slot(stream);
    a.paced = (p.ngrps == 1 && p.kgrps == 2 && p.sets % 8 == 0 && 2 * p.sets <= PACE_WGS && 2 * p.sets <= cu_count16() &&
               M * ldg * 4 >= tn_pace_min_bytes()) ? 1 : 0;
    a.partial = (float*)ws;
    a.bias_partial = gbias ? (float*)((char*)ws + hgv_align256((size_t)p.ngrps * p.kgrps * p.sets * T16_SET_FLOATS * 4)) : nullptr;
    if (f16) {
        const int rc = raise_lds16(k_gemm_tn16h);
        if (rc != HGV_OK) return rc;
        hipLaunchKernelGGL(k_gemm_tn16h, dim3((unsigned)(p.ngrps * p.kgrps * p.sets)), dim3(T16_THREADS), (size_t)T16H_LDS, stream, a);
    } else {
        const size_t lds = 2 * T16_BUF;
        const int rc = raise_lds16(k_gemm_tn16);
        if (rc != HGV_OK) return rc;
        hipLaunchKernelGGL(k_gemm_tn16, dim3((unsigned)(p.ngrps * p.kgrps * p.sets)), dim3(T16_THREADS), lds, stream, a);
    }
    r.GW[0] = GW1; r.ldgw[0] = ldgw1; r.GW[1] = GW2; r.ldgw[1] = ldgw2; r.gbias = gbias; r.accumulate = accumulate;
    const int64_t elems = (int64_t)p.ngrps * p.kgrps * T16_SET_FLOATS + (gbias ? (int64_t)p.ngrps * 9 * 256 : 0);
    hipLaunchKernelGGL(k_reduce_tn16, dim3((unsigned)hgv_cdiv(elems, 64)), dim3(256), 0, stream, r);
    HGV_CHECK_LAUNCH("gemm_tn16");
    return HGV_OK;
}

extern "C" int hgv_gemm_tn16(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                             const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias,
                             int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    return gemm_tn16_any(false, G, ldg, N, A1, lda1, K1, GW1, ldgw1, A2, lda2, K2, GW2, ldgw2, M, gbias, accumulate, ws, ws_bytes, stream);
}

// The same weight gradients from two scaled fp16 pieces per operand (three piece products); shapes, workspace and buffer
// rules of hgv_gemm_tn16.  Pad columns [N, ldg) / [K, lda) must be zero (they take part in the scale).
extern "C" int hgv_gemm_tn16h(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                              const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias,
                              int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    return gemm_tn16_any(true, G, ldg, N, A1, lda1, K1, GW1, ldgw1, A2, lda2, K2, GW2, ldgw2, M, gbias, accumulate, ws, ws_bytes, stream);
}

// One weight-gradient product for two gradient matrices of the same width N side by side: GW[2N, K] (+)= [Ga | Gb]^T A,
// gbias[2N] (+)= the column sums of Ga, then of Gb (N a multiple of 4; shapes as hgv_gemm_tn16_supported(M, 2N, K, 0)
// says).  See hgv_gemm_tn_gg for the use.
extern "C" int hgv_gemm_tn16_gg(const float* Ga, int64_t ldga, const float* Gb, int64_t ldgb, int N, const float* A, int64_t lda,
                                int K, int64_t M, float* GW, int64_t ldgw, float* gbias, int accumulate, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
    HGV_REQUIRE(Gb != nullptr && N > 0 && N % 4 == 0, HGV_EINVAL, "gemm_tn16_gg: two gradient matrices of a width that is a multiple of 4");
    return gemm_tn16_any(false, Ga, ldga, 2 * N, A, lda, K, GW, ldgw, nullptr, 0, 0, nullptr, 0, M, gbias, accumulate, ws, ws_bytes,
                         stream, Gb, ldgb, N);
}
