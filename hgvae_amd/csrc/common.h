// Shared helpers for the gfx950 kernels behind libhgvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define HGV_OK 0
#define HGV_EINVAL (-1)
#define HGV_EUNSUPPORTED (-2)
#define HGV_ELAUNCH (-3)
#define HGV_EWORKSPACE (-4)

// Thread-local message for the last failing call on this host thread.
void hgv_set_error(const char* fmt, ...);

#define HGV_REQUIRE(cond, code, ...)                                                               \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            hgv_set_error(__VA_ARGS__);                                                            \
            return (code);                                                                         \
        }                                                                                          \
    } while (0)

#define HGV_CHECK_LAUNCH(name)                                                                     \
    do {                                                                                           \
        hipError_t e__ = hipGetLastError();                                                        \
        if (e__ != hipSuccess) {                                                                   \
            hgv_set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));                \
            return HGV_ELAUNCH;                                                                    \
        }                                                                                          \
    } while (0)

static inline size_t hgv_align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t hgv_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool hgv_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// Launches that keep progress hints in a static device array (paced kernels) take one of HGV_PACE_SLOTS copies of it,
// chosen from the stream handle: launches that overlap on different streams then almost never share a copy (two handles
// hash to one slot with probability 1 / HGV_PACE_SLOTS; the hints only cost time when they do, never correctness).
#define HGV_PACE_SLOTS 8
static inline int hgv_pace_slot(hipStream_t stream) {
    const uintptr_t h = (uintptr_t)stream;
    return (int)(((h >> 4) ^ (h >> 9) ^ (h >> 15)) % HGV_PACE_SLOTS);
}

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on the XCD that
// also runs b+8).  This maps the hardware block id to a logical id such that the
// blocks sharing an XCD (and its private L2) cover one contiguous logical range.
// Bijective for every nb; affects speed only.
__device__ __forceinline__ int hgv_xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ int hgv_wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ float hgv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double hgv_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float hgv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Exclusive scan of one int per thread over a 256-thread block.
// lds must hold >= 5 ints.  Returns the exclusive prefix; *total gets the block sum.
__device__ __forceinline__ int hgv_block_excl_scan_256(int v, int* lds, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = hgv_wave_incl_scan(v, lane);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int t = lds[i];
        if (i < w) off += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011).
struct HgvPhilox {
    uint32_t k0, k1;
    __device__ __forceinline__ HgvPhilox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
        return rounds<10>(c0, c1, c2, c3);
    }
    // Philox4x32-R.  R = 7 is the smallest round count that passes BigCrush in the generator's paper;
    // keep decisions of dropout use it (4 integer multiplies per round dominate those kernels).
    template <int R>
    __device__ __forceinline__ uint4 rounds(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one: the integer multiplies
            // are quarter rate and all that these kernels wait for
            const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
            const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
            const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            const uint32_t n0 = hi1 ^ c1 ^ a, n2 = hi0 ^ c3 ^ b;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ float hgv_u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
