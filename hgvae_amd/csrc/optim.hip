// Adam update and gradient-norm clipping on flat fp32 buffers.
//
// Replaces torch.optim.Adam(model.parameters(), lr=...) (gvae/gvae_skipcon.py:294,
// autoencoder/vae_with_pool.py:270) and torch.nn.utils.clip_grad_norm_
// (autoencoder/vae_custom_pool.py:228-229) for models whose parameters and gradients live in one
// contiguous buffer each: the ~40 parameter tensors of these models (0.2-0.55 M floats) become one
// launch instead of several multi-tensor launches.  Arithmetic follows torch's single-tensor Adam:
//   g' = g * gscale (+ wd * p);  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace {

struct AdamArgs {
    float* p; const float* g; float* m; float* v; int64_t n;
    float lr, b1, b2, eps, wd;
    int64_t step; const int64_t* step_dev;     // step_dev (nullable) overrides step: graph replays advance it on the device
    const float* gscale;                       // nullable device scalar (clip coefficient)
};

__global__ __launch_bounds__(256) void k_adam(const AdamArgs a) {
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
        const double t = (double)(a.step_dev ? a.step_dev[0] : a.step);
        bc[0] = (float)(1.0 - pow((double)a.b1, t));
        bc[1] = (float)sqrt(1.0 - pow((double)a.b2, t));
    }
    __syncthreads();
    const float step_size = a.lr / bc[0], bc2s = bc[1];
    const float gs = a.gscale ? a.gscale[0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = a.p[i];
        float g = a.g[i] * gs;
        if (a.wd != 0.f) g += a.wd * p;
        const float m = a.b1 * a.m[i] + (1.f - a.b1) * g;
        const float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
        a.m[i] = m; a.v[i] = v;
        a.p[i] = p - step_size * (m / (sqrtf(v) / bc2s + a.eps));
    }
}

constexpr int NORM_BLOCKS = 512;

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = g[i];
        s += (double)x * (double)x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(64) void k_norm_finish(const double* __restrict__ partial, int nblocks, float max_norm,
                                                    float* __restrict__ norm_out, float* __restrict__ scale_out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s);
        norm_out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);        // torch: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
        scale_out[0] = c < 1.f ? c : 1.f;
    }
}

}  // namespace

extern "C" int hgv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, const int64_t* step_dev, const float* gscale_dev,
                             hipStream_t stream) {
    HGV_REQUIRE(n >= 0, HGV_EINVAL, "adam_step: negative size");
    if (n == 0) return HGV_OK;
    HGV_REQUIRE(p && g && m && v, HGV_EINVAL, "adam_step: null buffer");
    HGV_REQUIRE(step_dev || step >= 1, HGV_EINVAL, "adam_step: step counts from 1");
    HGV_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, HGV_EINVAL, "adam_step: bad hyper-parameters");
    AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, step_dev, gscale_dev};
    int64_t blocks = hgv_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    HGV_CHECK_LAUNCH("adam_step");
    return HGV_OK;
}

extern "C" size_t hgv_grad_norm_ws(void) { return hgv_align256(NORM_BLOCKS * sizeof(double)); }

// norm_out[0] = ||g||_2 ; scale_out[0] = min(1, max_norm / (norm + 1e-6)): the factor clip_grad_norm_ multiplies
// every gradient by (pass it to hgv_adam_step as gscale_dev).
extern "C" int hgv_grad_norm_clip(const float* g, int64_t n, float max_norm, float* norm_out, float* scale_out, void* ws,
                                  size_t ws_bytes, hipStream_t stream) {
    HGV_REQUIRE(n > 0 && g && norm_out && scale_out && ws, HGV_EINVAL, "grad_norm_clip: bad arguments");
    HGV_REQUIRE(ws_bytes >= hgv_grad_norm_ws() && hgv_aligned(ws, 256), HGV_EWORKSPACE, "grad_norm_clip: workspace too small or misaligned");
    int64_t blocks = hgv_cdiv(n, 256 * 4);
    if (blocks > NORM_BLOCKS) blocks = NORM_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sumsq_partial, dim3((unsigned)blocks), dim3(256), 0, stream, g, n, (double*)ws);
    hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(64), 0, stream, (const double*)ws, (int)blocks, max_norm, norm_out, scale_out);
    HGV_CHECK_LAUNCH("grad_norm_clip");
    return HGV_OK;
}
