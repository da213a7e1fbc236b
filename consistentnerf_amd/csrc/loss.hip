// Masked photometric / depth losses of ConsistentNeRF (V:1645-1648, V:1737, V:1786-1788, V:1865) and the
// vanilla MSE (R:769), forward value + gradient seeds for the compositing backward in one launch.
// Replaces 4 boolean-index gathers (each a host sync) per level.  One workgroup, fixed reduction order.
#include "common.hpp"

namespace {

constexpr int T = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < T / 64; ++i) s += sh[i];
  return s;
}

__global__ __launch_bounds__(T) void masked_loss_k(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                   const float* __restrict__ depth, const float* __restrict__ prior,
                                                   const float* __restrict__ mask, int64_t B, float far, float coef,
                                                   const float* __restrict__ counts, float g_scale,
                                                   float* __restrict__ loss, float* __restrict__ d_rgb,
                                                   float* __restrict__ d_depth) {
  __shared__ double sh[T / 64];
  // pass 1: counts and squared-error sums of the two sets (m==1, m==0; other values belong to neither)
  double n1 = 0, n0 = 0, s1 = 0, s0 = 0, sd = 0;
  for (int64_t i = threadIdx.x; i < B; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const bool in1 = m == 1.f, in0 = m == 0.f;
    float e = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = rgb[3 * i + c] - tgt[3 * i + c];
      e += d * d;
    }
    if (in1) { n1 += 1.0; s1 += (double)e; }
    if (in0) { n0 += 1.0; s0 += (double)e; }
    if (depth && in1) {
      const float d = depth[i] / far - prior[i] / far;
      sd += (double)(d * d);
    }
  }
  n1 = block_sum(n1, sh); n0 = block_sum(n0, sh);
  s1 = block_sum(s1, sh); s0 = block_sum(s0, sh); sd = block_sum(sd, sh);
  // global counts (e.g. all-reduced over ranks) override the local ones for the normalisation
  const double N1 = counts ? (double)counts[0] : n1;
  const double N0 = counts ? (double)counts[1] : n0;
  if (threadIdx.x == 0) {
    float l = (float)(s1 / (3.0 * N1));
    if (N0 > 0) l += coef * (float)(s0 / (3.0 * N0));
    loss[0] = l;
    loss[1] = depth ? (float)(sd / N1) : 0.f;
  }
  const float w1 = g_scale * (float)(2.0 / (3.0 * N1));
  const float w0 = N0 > 0 ? g_scale * coef * (float)(2.0 / (3.0 * N0)) : 0.f;
  const float wd = g_scale * (float)(2.0 / N1) / far;
  for (int64_t i = threadIdx.x; i < B; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const float w = m == 1.f ? w1 : (m == 0.f ? w0 : 0.f);
    if (d_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d_rgb[3 * i + c] = w * (rgb[3 * i + c] - tgt[3 * i + c]);
    }
    if (d_depth) d_depth[i] = (depth && m == 1.f) ? wd * (depth[i] / far - prior[i] / far) : 0.f;
  }
}

}  // namespace

extern "C" int64_t cnerf_loss_ws_floats(void) { return 0; }

extern "C" int cnerf_masked_loss(const float* rgb, const float* target, const float* depth, const float* prior,
                                 const float* mask, int64_t B, float far, float coef, const float* counts,
                                 float g_scale, float* loss, float* d_rgb, float* d_depth, float* workspace,
                                 void* stream) {
  (void)workspace;
  if (!rgb || !target || !loss || B <= 0 || (depth && !prior) || !(far > 0.f)) return CNERF_E_ARG;
  hipLaunchKernelGGL(masked_loss_k, dim3(1), dim3(T), 0, cn_stream(stream), rgb, target, depth, prior, mask, B, far,
                     coef, counts, g_scale, loss, d_rgb, d_depth);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
