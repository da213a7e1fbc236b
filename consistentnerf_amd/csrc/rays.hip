// Ray generation and ray-batch assembly: get_rays (H:164-173), ndc_rays (H:186-202) and the
// [B, 8|11] pack of render() (R:100-125).  One thread per ray; compiled without FMA contraction so
// every product/sum rounds like the reference's separate ATen ops.
#include "common.hpp"

namespace {

struct Cam { float r[9]; float t[3]; };   // c2w rotation (row-major) and translation

__device__ __forceinline__ void finish_ray(float ox, float oy, float oz, float dx, float dy, float dz, float near,
                                           float far, int vd, int ndc, float ax, float ay, float* __restrict__ out,
                                           int rs) {
  float vx = 0.f, vy = 0.f, vz = 0.f;
  if (vd) {   // viewdirs from the PRE-NDC direction (R:103-110)
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    vx = dx / n; vy = dy / n; vz = dz / n;
  }
  if (ndc) {  // H:188-202 with near plane 1
    const float t = -(1.f + oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    const float o0 = ax * ox / oz;
    const float o1 = ay * oy / oz;
    const float o2 = 1.f + 2.f / oz;
    const float d0 = ax * (dx / dz - ox / oz);
    const float d1 = ay * (dy / dz - oy / oz);
    const float d2 = -2.f / oz;
    ox = o0; oy = o1; oz = o2; dx = d0; dy = d1; dz = d2;
  }
  out[0] = ox; out[1] = oy; out[2] = oz; out[3] = dx; out[4] = dy; out[5] = dz; out[6] = near; out[7] = far;
  if (vd) { out[8] = vx; out[9] = vy; out[10] = vz; }
}

__global__ void gen_rays_k(int H, int W, float fx, float fy, float cx, float cy, Cam c, float near, float far, int vd,
                           int ndc, float ax, float ay, float* __restrict__ rays) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)H * W) return;
  const int j = (int)(idx / W), i = (int)(idx - (int64_t)j * W);
  const float d0 = ((float)i - cx) / fx, d1 = -((float)j - cy) / fy, d2 = -1.f;   // H:167
  // rays_d = sum(dirs[..., None, :] * c2w[:3,:3], -1)  (H:170): three products, then a 3-term sum
  const float dx = d0 * c.r[0] + d1 * c.r[1] + d2 * c.r[2];
  const float dy = d0 * c.r[3] + d1 * c.r[4] + d2 * c.r[5];
  const float dz = d0 * c.r[6] + d1 * c.r[7] + d2 * c.r[8];
  const int rs = vd ? 11 : 8;
  finish_ray(c.t[0], c.t[1], c.t[2], dx, dy, dz, near, far, vd, ndc, ax, ay, rays + idx * rs, rs);
}

__global__ void pack_rays_k(const float* __restrict__ ro, const float* __restrict__ rd, int64_t B, float near,
                            float far, int vd, int ndc, float ax, float ay, float* __restrict__ rays) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B) return;
  const int rs = vd ? 11 : 8;
  finish_ray(ro[idx * 3], ro[idx * 3 + 1], ro[idx * 3 + 2], rd[idx * 3], rd[idx * 3 + 1], rd[idx * 3 + 2], near, far,
             vd, ndc, ax, ay, rays + idx * rs, rs);
}

}  // namespace

extern "C" int cnerf_gen_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host, float near,
                              float far, int use_viewdirs, int ndc, float ndc_ax, float ndc_ay, float* rays,
                              void* stream) {
  if (!c2w_host || !rays || H <= 0 || W <= 0) return CNERF_E_ARG;
  Cam c;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) c.r[3 * r + k] = c2w_host[4 * r + k];
    c.t[r] = c2w_host[4 * r + 3];
  }
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(gen_rays_k, dim3((unsigned)cn_div_up(n, 256)), dim3(256), 0, cn_stream(stream), H, W, fx, fy, cx,
                     cy, c, near, far, use_viewdirs, ndc, ndc_ax, ndc_ay, rays);
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
