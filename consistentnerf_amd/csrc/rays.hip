// Ray generation and ray-batch assembly: get_rays (H:164-173), ndc_rays (H:186-202) and the
// [B, 8|11] pack of render() (R:100-125).  One thread per ray; compiled without FMA contraction so
// every product/sum rounds like the reference's separate ATen ops.
#include "raygen.hpp"

namespace {

__device__ __forceinline__ void put_ray(const float (&o)[3], const float (&d)[3], const float (&v)[3], float near, float far,
                                        int vd, float* __restrict__ out) {
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2]; out[6] = near; out[7] = far;
  if (vd) { out[8] = v[0]; out[9] = v[1]; out[10] = v[2]; }
}

__global__ void gen_rays_k(int64_t n, RayGenDev g, float* __restrict__ rays) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float o[3], d[3], v[3];
  cn_gen_ray(g, g.first + idx, o, d, v);
  put_ray(o, d, v, g.near, g.far, g.vd, rays + idx * (g.vd ? 11 : 8));
}

__global__ void pack_rays_k(const float* __restrict__ ro, const float* __restrict__ rd, int64_t B, float near,
                            float far, int vd, int ndc, float ax, float ay, float* __restrict__ rays) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B) return;
  float o[3], d[3], v[3];
  cn_finish_ray(ro[idx * 3], ro[idx * 3 + 1], ro[idx * 3 + 2], rd[idx * 3], rd[idx * 3 + 1], rd[idx * 3 + 2], vd, ndc, ax, ay,
                o, d, v);
  put_ray(o, d, v, near, far, vd, rays + idx * (vd ? 11 : 8));
}

}  // namespace

extern "C" int cnerf_gen_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host, float near,
                              float far, int use_viewdirs, int ndc, float ndc_ax, float ndc_ay, float* rays,
                              void* stream) {
  if (!c2w_host || !rays || H <= 0 || W <= 0) return CNERF_E_ARG;
  cnerf_raygen c;
  c.H = H; c.W = W; c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy;
  for (int k = 0; k < 12; ++k) c.c2w[k] = c2w_host[k];
  c.near = near; c.far = far; c.use_viewdirs = use_viewdirs; c.ndc = ndc; c.ndc_ax = ndc_ax; c.ndc_ay = ndc_ay; c.first = 0;
  RayGenDev g;
  int rc = cn_make_raygen(&c, &g);
  if (rc) return rc;
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gen_rays_k, dim3((unsigned)cn_div_up(n, 256)), dim3(256), 0, cn_stream(stream), n, g, rays);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_pack_rays(const float* rays_o, const float* rays_d, int64_t B, float near, float far,
                               int use_viewdirs, int ndc, float ndc_ax, float ndc_ay, float* rays, void* stream) {
  if (!rays_o || !rays_d || !rays || B < 0) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  hipLaunchKernelGGL(pack_rays_k, dim3((unsigned)cn_div_up(B, 256)), dim3(256), 0, cn_stream(stream), rays_o, rays_d,
                     B, near, far, use_viewdirs, ndc, ndc_ax, ndc_ay, rays);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
