// Reparameterisation, KL divergence and mean-squared reconstruction error, forward and backward.
//
// Reference: VAE.reparameterize gvae/gvae_skipcon.py:138-142 (mean over n_samples draws of
// mean + exp(log_std) * eps), VAE.kl_divergence :153-156, F.mse_loss :174.
// mean and log_std are the two halves of one [N, 2Z] head output (chunk(2, dim=-1) at :118),
// so they are addressed as columns [0,Z) and [Z,2Z) of one buffer with leading dimension ldm.
//
// Scalar reductions are two-stage (per-workgroup partial -> one finishing workgroup in fp64),
// no float atomics, so losses are bitwise reproducible.
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

// z = mean_k(mean + std * eps_k); eps explicit [n_samples, N, Z] or Philox.  Also writes
// eps_bar = mean_k eps_k (needed by the backward pass).
__global__ __launch_bounds__(256) void k_reparam_fwd(const float* __restrict__ ml, int64_t ldm, int64_t N, int Z,
                                                     int n_samples, const float* __restrict__ eps, uint64_t seed,
                                                     uint64_t offset0, const uint64_t* __restrict__ offset_dev,
                                                     float* __restrict__ z, int64_t ldz,
                                                     float* __restrict__ eps_bar) {
    const int64_t total = N * Z;
    const float inv_n = 1.0f / (float)n_samples;
    const uint64_t offset = offset0 + (offset_dev ? offset_dev[0] : 0ull);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Z;
        const int c = (int)(i - row * Z);
        const float mu = ml[row * ldm + c];
        const float sd = expf(ml[row * ldm + Z + c]);
        float acc = 0.f, eacc = 0.f;
        if (eps) {
            for (int k = 0; k < n_samples; ++k) {
                const float e = eps[(int64_t)k * total + i];
                acc += mu + sd * e;     // same order as the reference's stack().mean(0)
                eacc += e;
            }
        } else {
            const HgvPhilox ph(seed);
            for (int k = 0; k < n_samples; k += 4) {
                const uint4 r = ph((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)offset + (uint32_t)(k >> 2),
                                   (uint32_t)(offset >> 32) ^ 0x5EEDu);
                const float2 a = box_muller(r.x, r.y), b = box_muller(r.z, r.w);
                const float e4[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < n_samples) { acc += mu + sd * e4[j]; eacc += e4[j]; }
            }
        }
        z[row * ldz + c] = acc * inv_n;
        eps_bar[i] = eacc * inv_n;
    }
}

// g_mean = gz ; g_log_std = gz * std * eps_bar
__global__ __launch_bounds__(256) void k_reparam_bwd(const float* __restrict__ gz, int64_t ldgz,
                                                     const float* __restrict__ ml, int64_t ldm, int64_t N, int Z,
                                                     const float* __restrict__ eps_bar, float* __restrict__ gml,
                                                     int64_t ldg) {
    const int64_t total = N * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Z;
        const int c = (int)(i - row * Z);
        const float g = gz[row * ldgz + c];
        gml[row * ldg + c] = g;
        gml[row * ldg + Z + c] = g * expf(ml[row * ldm + Z + c]) * eps_bar[i];
    }
}

__device__ __forceinline__ void block_reduce_store(float v, float* partial) {
    __shared__ float sh[4];
    v = hgv_wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_kl_partial(const float* __restrict__ ml, int64_t ldm, int64_t N, int Z,
                                                    float* __restrict__ partial) {
    const int64_t total = N * Z;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Z;
        const int c = (int)(i - row * Z);
        const float mu = ml[row * ldm + c], ls = ml[row * ldm + Z + c];
        const float sd = expf(ls);
        acc += 1.f + 2.f * ls - mu * mu - sd * sd;
    }
    block_reduce_store(acc, partial);
}

__global__ __launch_bounds__(256) void k_mse_partial(const float* __restrict__ r, int64_t ldr,
                                                     const float* __restrict__ x, int64_t ldx, int64_t N, int F,
                                                     float* __restrict__ partial) {
    const int64_t total = N * F;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / F;
        const int c = (int)(i - row * F);
        const float d = r[row * ldr + c] - x[row * ldx + c];
        acc += d * d;
    }
    block_reduce_store(acc, partial);
}

__global__ __launch_bounds__(256) void k_finish_sum(const float* __restrict__ partial, int n, double mult,
                                                    float* __restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += (double)partial[i];
    v = hgv_wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((sh[0] + sh[1] + sh[2] + sh[3]) * mult);
}

// d KL / d mean = mu / N ; d KL / d log_std = (exp(2 ls) - 1) / N, times the upstream scalar *gout
__global__ __launch_bounds__(256) void k_kl_bwd(const float* __restrict__ ml, int64_t ldm, int64_t N, int Z,
                                                const float* __restrict__ gout, float* __restrict__ gml,
                                                int64_t ldg, int accumulate) {
    const int64_t total = N * Z;
    const float s = gout[0] / (float)N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Z;
        const int c = (int)(i - row * Z);
        const float mu = ml[row * ldm + c], ls = ml[row * ldm + Z + c];
        const float sd = expf(ls);
        const float gm = s * mu, gl = s * (sd * sd - 1.f);
        if (accumulate) { gml[row * ldg + c] += gm; gml[row * ldg + Z + c] += gl; }
        else { gml[row * ldg + c] = gm; gml[row * ldg + Z + c] = gl; }
    }
}

__global__ __launch_bounds__(256) void k_mse_bwd(const float* __restrict__ r, int64_t ldr, const float* __restrict__ x,
                                                 int64_t ldx, int64_t N, int F, const float* __restrict__ gout,
                                                 float* __restrict__ gr, int64_t ldg) {
    const int64_t total = N * F;
    const float s = 2.0f * gout[0] / ((float)N * (float)F);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / F;
        const int c = (int)(i - row * F);
        gr[row * ldg + c] = s * (r[row * ldr + c] - x[row * ldx + c]);
    }
}

int grid_for(int64_t total) {
    int64_t g = hgv_cdiv(total, 256);
    if (g > RED_BLOCKS) g = RED_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" size_t hgv_loss_ws(void) { return RED_BLOCKS * sizeof(float); }

extern "C" int hgv_reparam_fwd(const float* ml, int64_t ldm, int64_t N, int Z, int n_samples, const float* eps,
                               uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float* z, int64_t ldz,
                               float* eps_bar,
                               hipStream_t stream) {
    HGV_REQUIRE(N >= 0 && Z > 0 && n_samples > 0, HGV_EINVAL, "reparam_fwd: bad sizes");
    if (N == 0) return HGV_OK;
    HGV_REQUIRE(ml && z && eps_bar && ldm >= 2 * Z && ldz >= Z, HGV_EINVAL, "reparam_fwd: null/short buffer");
    hipLaunchKernelGGL(k_reparam_fwd, dim3(grid_for(N * Z)), dim3(256), 0, stream, ml, ldm, N, Z, n_samples, eps, seed,
                       offset, offset_dev, z, ldz, eps_bar);
    HGV_CHECK_LAUNCH("reparam_fwd");
    return HGV_OK;
}

extern "C" int hgv_reparam_bwd(const float* gz, int64_t ldgz, const float* ml, int64_t ldm, int64_t N, int Z,
                               const float* eps_bar, float* gml, int64_t ldg, hipStream_t stream) {
    HGV_REQUIRE(N >= 0 && Z > 0, HGV_EINVAL, "reparam_bwd: bad sizes");
    if (N == 0) return HGV_OK;
    HGV_REQUIRE(gz && ml && eps_bar && gml && ldm >= 2 * Z && ldg >= 2 * Z && ldgz >= Z, HGV_EINVAL, "reparam_bwd: null/short buffer");
    hipLaunchKernelGGL(k_reparam_bwd, dim3(grid_for(N * Z)), dim3(256), 0, stream, gz, ldgz, ml, ldm, N, Z, eps_bar, gml, ldg);
    HGV_CHECK_LAUNCH("reparam_bwd");
    return HGV_OK;
}

// out[0] = mean_i( -0.5 * sum_d (1 + 2 ls - mu^2 - exp(ls)^2) )
extern "C" int hgv_kl_fwd(const float* ml, int64_t ldm, int64_t N, int Z, float* out, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
    HGV_REQUIRE(N > 0 && Z > 0 && ml && out && ws && ldm >= 2 * Z, HGV_EINVAL, "kl_fwd: bad arguments");
    HGV_REQUIRE(ws_bytes >= hgv_loss_ws(), HGV_EWORKSPACE, "kl_fwd: workspace too small");
    const int g = grid_for(N * Z);
    hipLaunchKernelGGL(k_kl_partial, dim3(g), dim3(256), 0, stream, ml, ldm, N, Z, (float*)ws);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, stream, (const float*)ws, g, -0.5 / (double)N, out);
    HGV_CHECK_LAUNCH("kl_fwd");
    return HGV_OK;
}

extern "C" int hgv_kl_bwd(const float* ml, int64_t ldm, int64_t N, int Z, const float* gout, float* gml, int64_t ldg,
                          int accumulate, hipStream_t stream) {
    HGV_REQUIRE(N > 0 && Z > 0 && ml && gout && gml && ldm >= 2 * Z && ldg >= 2 * Z, HGV_EINVAL, "kl_bwd: bad arguments");
    hipLaunchKernelGGL(k_kl_bwd, dim3(grid_for(N * Z)), dim3(256), 0, stream, ml, ldm, N, Z, gout, gml, ldg, accumulate);
    HGV_CHECK_LAUNCH("kl_bwd");
    return HGV_OK;
}

// out[0] = mean over N*F of (r - x)^2
extern "C" int hgv_mse_fwd(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F, float* out,
                           void* ws, size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(N > 0 && F > 0 && r && x && out && ws && ldr >= F && ldx >= F, HGV_EINVAL, "mse_fwd: bad arguments");
    HGV_REQUIRE(ws_bytes >= hgv_loss_ws(), HGV_EWORKSPACE, "mse_fwd: workspace too small");
    const int g = grid_for(N * F);
    hipLaunchKernelGGL(k_mse_partial, dim3(g), dim3(256), 0, stream, r, ldr, x, ldx, N, F, (float*)ws);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, stream, (const float*)ws, g, 1.0 / ((double)N * (double)F), out);
    HGV_CHECK_LAUNCH("mse_fwd");
    return HGV_OK;
}

extern "C" int hgv_mse_bwd(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F,
                           const float* gout, float* gr, int64_t ldg, hipStream_t stream) {
    HGV_REQUIRE(N > 0 && F > 0 && r && x && gout && gr && ldr >= F && ldx >= F && ldg >= F, HGV_EINVAL, "mse_bwd: bad arguments");
    hipLaunchKernelGGL(k_mse_bwd, dim3(grid_for(N * F)), dim3(256), 0, stream, r, ldr, x, ldx, N, F, gout, gr, ldg);
    HGV_CHECK_LAUNCH("mse_bwd");
    return HGV_OK;
}
