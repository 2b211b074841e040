// Memory-side model of the NT projection (no products): 256 workgroups of W waves walk 16-row tiles of two row-major
// matrices (ld floats per row) and write one.  PATTERN 0: the operand layout of the matrix cores (lane = row + 16 * piece,
// two 16-byte loads per 32-column step, ring of 4 steps); 1: the tile's 16 rows as one contiguous range, 16 bytes per lane
// and load, consecutive lanes consecutive addresses.  Build: hipcc --offload-arch=gfx950 -O3 -o nt_mem_lab nt_mem_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PATTERN, int NSRC>
__global__ __launch_bounds__(1024) void k_walk(const float* A1, const float* A2, float* C, long M, int ld, long ntiles) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const long gw = (long)wib * gridDim.x + blockIdx.x, nw = (long)wpb * gridDim.x;
    const int tile_f4 = 16 * ld / 4;                     // float4 per tile
    for (long tile = gw; tile < ntiles; tile += nw) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (PATTERN == 0) {
            const int i = lane & 15, kq = lane >> 4;
            const long row = tile * 16 + i;
            const int steps = (ld * 4 + 127) / 128;
#pragma unroll 1
            for (int src = 0; src < NSRC; ++src) {
                const char* p = reinterpret_cast<const char*>((src ? A2 : A1) + row * ld);
                for (int s0 = 0; s0 < steps; s0 += 4) {
                    f32x4 v[4][2];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        int off = (s0 + d) * 128 + kq * 32;
                        if (off > ld * 4 - 32) off = ld * 4 - 32;
                        v[d][0] = *reinterpret_cast<const f32x4*>(p + off);
                        v[d][1] = *reinterpret_cast<const f32x4*>(p + off + 16);
                    }
#pragma unroll
                    for (int d = 0; d < 4; ++d) acc += v[d][0] * v[d][1];
                }
            }
        } else {
            const int nload = (tile_f4 + 63) / 64;
#pragma unroll 1
            for (int src = 0; src < NSRC; ++src) {
                const f32x4* p = reinterpret_cast<const f32x4*>((src ? A2 : A1) + tile * 16 * ld);
                for (int j0 = 0; j0 < nload; j0 += 3) {
                    f32x4 v[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        int e = (j0 + d) * 64 + lane;
                        if (e > tile_f4 - 1) e = tile_f4 - 1;
                        v[d] = p[e];
                    }
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc += v[d];
                }
            }
        }
        // store: PATTERN 0 as the product does (lane (i, kq): 16 bytes at column 16 n + 4 kq of row i), 1 contiguous
        if (PATTERN == 0) {
            const int i = lane & 15, kq = lane >> 4;
            float* c = C + (tile * 16 + i) * ld;
            for (int n = 0; n * 16 + 4 * kq < ld; ++n) *reinterpret_cast<f32x4*>(c + n * 16 + 4 * kq) = acc;
        } else {
            f32x4* c = reinterpret_cast<f32x4*>(C + tile * 16 * ld);
            for (int e = lane; e < tile_f4; e += 64) c[e] = acc;
        }
    }
}

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 1262000;
    const int ld = 140;
    const long ntiles = M / 16;
    float *A1, *A2, *C;
    CK(hipMalloc(&A1, M * ld * 4)); CK(hipMalloc(&A2, M * ld * 4)); CK(hipMalloc(&C, M * ld * 4));
    CK(hipMemset(A1, 0, M * ld * 4)); CK(hipMemset(A2, 0, M * ld * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nsrc = 1; nsrc <= 2; ++nsrc)
        for (int pattern = 0; pattern < 2; ++pattern)
            for (int waves : {4, 8, 12, 16}) {
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    if (pattern == 0 && nsrc == 1) hipLaunchKernelGGL((k_walk<0, 1>), dim3(256), dim3(64 * waves), 0, 0, A1, A2, C, M, ld, ntiles);
                    if (pattern == 0 && nsrc == 2) hipLaunchKernelGGL((k_walk<0, 2>), dim3(256), dim3(64 * waves), 0, 0, A1, A2, C, M, ld, ntiles);
                    if (pattern == 1 && nsrc == 1) hipLaunchKernelGGL((k_walk<1, 1>), dim3(256), dim3(64 * waves), 0, 0, A1, A2, C, M, ld, ntiles);
                    if (pattern == 1 && nsrc == 2) hipLaunchKernelGGL((k_walk<1, 2>), dim3(256), dim3(64 * waves), 0, 0, A1, A2, C, M, ld, ntiles);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                const double gb = (double)M * ld * 4 * (nsrc + 1) / 1e9;
                printf("sources %d pattern %d waves %2d: %7.1f us  %5.2f TB/s\n", nsrc, pattern, waves, best * 1e3, gb / best);
            }
    return 0;
}
