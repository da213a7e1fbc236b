// Optimiser tail in one launch over a flat parameter buffer: clip_grad_value_ (V:1983) + Adam with
// betas (0.9, 0.999), eps 1e-8 (R:210, R:780); the caller passes the decayed lr of R:784-788.
// Arithmetic follows torch.optim.Adam's single-tensor path (lerp / addcmul / sqrt / addcdiv).
#include <math.h>

#include "common.hpp"

namespace {
__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float w1, float beta2, float w2, float step_size,
                       float bc2_sqrt, float eps, float clip, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float mi = m[i] + w1 * (gi - m[i]);               // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + w2 * (gi * gi);         // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / denom);                 // param.addcdiv_(exp_avg, denom, -step_size)
  }
}
// the same update with every scalar read from device memory: hyp[8] = {1-beta1, beta2, 1-beta2, lr / bc1, sqrt(bc2), eps, clip,
// grad_scale}.  Nothing of the step is baked into the launch, so a captured hipGraph of the training step can be replayed
// while the host rewrites the (pinned) source of hyp between replays.
__global__ void adam_dev_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, int64_t n, const float* __restrict__ hyp) {
  const float w1 = hyp[0], beta2 = hyp[1], w2 = hyp[2], step_size = hyp[3], bc2_sqrt = hyp[4], eps = hyp[5], clip = hyp[6],
              gscale = hyp[7];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * beta2 + w2 * (gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / denom);
  }
}
}  // namespace

extern "C" int cnerf_adam_hyper(int step, double lr, double beta1, double beta2, double eps, float clip, float grad_scale,
                                float* hyp8_host) {
  if (!hyp8_host || step < 1) return CNERF_E_ARG;
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hyp8_host[0] = (float)(1.0 - beta1); hyp8_host[1] = (float)beta2; hyp8_host[2] = (float)(1.0 - beta2);
  hyp8_host[3] = (float)(lr / bc1); hyp8_host[4] = (float)sqrt(bc2); hyp8_host[5] = (float)eps;
  hyp8_host[6] = clip; hyp8_host[7] = grad_scale;
  return CNERF_OK;
}

extern "C" int cnerf_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyp8_dev,
                                   void* stream) {
  if (!p || !g || !m || !v || !hyp8_dev || n < 0) return CNERF_E_ARG;
  if (n == 0) return CNERF_OK;
  int64_t blocks = cn_div_up(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_dev_k, dim3((unsigned)blocks), dim3(256), 0, cn_stream(stream), p, g, m, v, n, hyp8_dev);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                               double beta1, double beta2, double eps, float clip, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return CNERF_E_ARG;
  if (n == 0) return CNERF_OK;
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  int64_t blocks = cn_div_up(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_k, dim3((unsigned)blocks), dim3(256), 0, cn_stream(stream), p, g, m, v, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps,
                     clip, grad_scale);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
