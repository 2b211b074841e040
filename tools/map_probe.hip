// Copy of an [N x 140] float matrix (138 live columns = 35 16-byte chunks per row) with three thread mappings,
// to see what the mapping of the epilogue kernels costs.  hipcc -O3 --offload-arch=gfx950 tools/map_probe.hip -o tools/bin/map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
constexpr int CH = 35, CP = 18;
// A: thread = (row slot, chunk pair cp): chunks cp and cp + 18 of one row (the epilogue kernels' mapping)
__global__ __launch_bounds__(256) void k_a(const float4* __restrict__ x, float4* __restrict__ y, int64_t N, int64_t rpb) {
    const int t = threadIdx.x, RPB = 256 / CP;
    if (t >= RPB * CP) return;
    const int rsub = t / CP, cp = t - rsub * CP;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    for (int64_t row = r0 + rsub; row < r1; row += RPB) {
        float4 a = x[row * CH + cp], b = x[row * CH + min(cp + CP, CH - 1)];
        a.x *= 2.f; b.x *= 2.f;
        y[row * CH + cp] = a;
        if (cp + CP < CH) y[row * CH + cp + CP] = b;
    }
}
// B: thread = (row pair slot, chunk c): chunk c of rows 2k and 2k + 1
__global__ __launch_bounds__(256) void k_b(const float4* __restrict__ x, float4* __restrict__ y, int64_t N, int64_t rpb) {
    const int t = threadIdx.x, RPB = 256 / CH;
    if (t >= RPB * CH) return;
    const int rsub = t / CH, c = t - rsub * CH;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);      // rpb even
    for (int64_t row = r0 + 2 * rsub; row < r1; row += 2 * RPB) {
        const int64_t rb = min(row + 1, N - 1);
        float4 a = x[row * CH + c], b = x[rb * CH + c];
        a.x *= 2.f; b.x *= 2.f;
        y[row * CH + c] = a;
        if (row + 1 < r1) y[rb * CH + c] = b;
    }
}
// C: thread = (row slot, chunk c), rows r and r + RPB: both instructions cover contiguous ranges
__global__ __launch_bounds__(256) void k_c(const float4* __restrict__ x, float4* __restrict__ y, int64_t N, int64_t rpb) {
    const int t = threadIdx.x, RPB = 256 / CH;
    if (t >= RPB * CH) return;
    const int rsub = t / CH, c = t - rsub * CH;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    for (int64_t row = r0 + rsub; row < r1; row += 2 * RPB) {
        const int64_t rb = min(row + RPB, N - 1);
        float4 a = x[row * CH + c], b = x[rb * CH + c];
        a.x *= 2.f; b.x *= 2.f;
        y[row * CH + c] = a;
        if (row + RPB < r1) y[rb * CH + c] = b;
    }
}
// E: mapping A plus W dependent multiply-adds per loaded value (how much arithmetic the copy tolerates)
template <int W>
__global__ __launch_bounds__(256) void k_e(const float4* __restrict__ x, float4* __restrict__ y, int64_t N, int64_t rpb, float s) {
    const int t = threadIdx.x, RPB = 256 / CP;
    if (t >= RPB * CP) return;
    const int rsub = t / CP, cp = t - rsub * CP;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    for (int64_t row = r0 + rsub; row < r1; row += RPB) {
        float4 a = x[row * CH + cp], b = x[row * CH + min(cp + CP, CH - 1)];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            a.x = a.x * s + 1.f; a.y = a.y * s + 1.f; a.z = a.z * s + 1.f; a.w = a.w * s + 1.f;
            b.x = b.x * s + 1.f; b.y = b.y * s + 1.f; b.z = b.z * s + 1.f; b.w = b.w * s + 1.f;
        }
        y[row * CH + cp] = a;
        if (cp + CP < CH) y[row * CH + cp + CP] = b;
    }
}
// F: E with R rows per iteration, all loads issued before any arithmetic
template <int W, int R>
__global__ __launch_bounds__(256) void k_f(const float4* __restrict__ x, float4* __restrict__ y, int64_t N, int64_t rpb, float s) {
    const int t = threadIdx.x, RPB = 256 / CP;
    if (t >= RPB * CP) return;
    const int rsub = t / CP, cp = t - rsub * CP;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    for (int64_t row = r0 + rsub; row < r1; row += R * RPB) {
        float4 a[R], b[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = min(row + u * RPB, r1 - 1);
            a[u] = x[r * CH + cp]; b[u] = x[r * CH + min(cp + CP, CH - 1)];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                a[u].x = a[u].x * s + 1.f; a[u].y = a[u].y * s + 1.f; a[u].z = a[u].z * s + 1.f; a[u].w = a[u].w * s + 1.f;
                b[u].x = b[u].x * s + 1.f; b[u].y = b[u].y * s + 1.f; b[u].z = b[u].z * s + 1.f; b[u].w = b[u].w * s + 1.f;
            }
            const int64_t r = row + u * RPB;
            if (r < r1) {
                y[r * CH + cp] = a[u];
                if (cp + CP < CH) y[r * CH + cp + CP] = b[u];
            }
        }
    }
}
// D: flat
__global__ __launch_bounds__(256) void k_d(const float4* __restrict__ x, float4* __restrict__ y, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float4 a = x[i]; a.x *= 2.f; y[i] = a;
    }
}
int main() {
    const int64_t N = 160984;
    float4 *x, *y;
    hipMalloc(&x, N * CH * 16); hipMalloc(&y, N * CH * 16);
    hipMemset(x, getenv("FILL") ? 0x3f : 0, N * CH * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nb = 1024; const int64_t rpb = ((N + nb - 1) / nb + 1) / 2 * 2;
    const int nblk = (int)((N + rpb - 1) / rpb);
    for (int k = 0; k < 9; ++k) {
        float best = 1e9f;
        for (int it = 0; it < 30; ++it) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(k_a, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb);
            if (k == 1) hipLaunchKernelGGL(k_b, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb);
            if (k == 2) hipLaunchKernelGGL(k_c, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb);
            if (k == 3) hipLaunchKernelGGL(k_d, dim3(4096), dim3(256), 0, 0, x, y, N * CH);
            if (k == 4) hipLaunchKernelGGL(k_e<4>, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb, 0.5f);
            if (k == 5) hipLaunchKernelGGL(k_e<8>, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb, 0.5f);
            if (k == 6) hipLaunchKernelGGL(k_e<16>, dim3(nblk), dim3(256), 0, 0, x, y, N, rpb, 0.5f);
            if (k == 7) hipLaunchKernelGGL((k_f<8, 2>), dim3(nblk), dim3(256), 0, 0, x, y, N, rpb, 0.5f);
            if (k == 8) hipLaunchKernelGGL((k_f<8, 4>), dim3(nblk), dim3(256), 0, 0, x, y, N, rpb, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it > 3 && ms < best) best = ms;
        }
        printf("mapping %c%s: %.1f us  %.2f TB/s\n", 'A' + (k < 4 ? k : 4), k == 4 ? " +64 VALU" : k == 5 ? " +128 VALU" : k == 6 ? " +256 VALU" : k == 7 ? " +128 VALU, 2 rows in flight" : k == 8 ? " +128 VALU, 4 rows in flight" : "",
               best * 1e3, 2.0 * N * CH * 16 / best / 1e9);
    }
    return 0;
}
