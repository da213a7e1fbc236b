// Rays of a pinhole camera generated where they are consumed ("for C5 generate rays in-kernel from c2w
// instead of materialising 33 MB per frame"): get_rays H:164-173, viewdirs from the pre-NDC direction R:103-110,
// ndc_rays H:186-202 with near plane 1.  The one definition of this arithmetic: gen_rays_k / pack_rays_k (rays.hip)
// write its results to HBM, the *_cam paths of coarse_z / mlp_fwd / composite evaluate it per consumer — bit-identical
// by construction.  Compiled without FMA contraction (every product / sum rounds like the reference's separate ATen ops).
#pragma once
#include "common.hpp"

struct RayGenDev {
  int on, W;                 // on == 0: rays come from memory
  float fx, fy, cx, cy;
  float r[9], t[3];          // c2w rotation (row-major) and translation
  float near, far;
  int vd, ndc;
  float ax, ay;              // the two NDC coefficients of H:193-199
  int64_t first;             // row-major pixel index of ray 0 of the call
};

// o, d after the optional NDC warp; v = d / |d| of the PRE-NDC direction (only when vd)
__device__ __forceinline__ void cn_finish_ray(float ox, float oy, float oz, float dx, float dy, float dz, int vd, int ndc,
                                              float ax, float ay, float (&o)[3], float (&d)[3], float (&v)[3]) {
  v[0] = v[1] = v[2] = 0.f;
  if (vd) {   // viewdirs from the PRE-NDC direction (R:103-110)
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    v[0] = dx / n; v[1] = dy / n; v[2] = dz / n;
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
  o[0] = ox; o[1] = oy; o[2] = oz; d[0] = dx; d[1] = dy; d[2] = dz;
}

// pre-NDC direction of pixel (row j, column i) of the camera (H:167-170).  Works on any address space of `g`.
template <class G>
__device__ __forceinline__ void cn_raw_dir(const G& g, int j, int i, float& dx, float& dy, float& dz) {
  const float d0 = ((float)i - g.cx) / g.fx, d1 = -((float)j - g.cy) / g.fy, d2 = -1.f;   // H:167
  // rays_d = sum(dirs[..., None, :] * c2w[:3,:3], -1)  (H:170): three products, then a 3-term sum
  dx = d0 * g.r[0] + d1 * g.r[1] + d2 * g.r[2];
  dy = d0 * g.r[3] + d1 * g.r[4] + d2 * g.r[5];
  dz = d0 * g.r[6] + d1 * g.r[7] + d2 * g.r[8];
}

// ray `idx` (row-major pixel) of the camera.  Works on any address space of `g` (kernarg segment included).
template <class G>
__device__ __forceinline__ void cn_gen_ray(const G& g, int64_t idx, float (&o)[3], float (&d)[3], float (&v)[3]) {
  const int W = g.W;
  const int j = (int)(idx / W), i = (int)(idx - (int64_t)j * W);
  float dx, dy, dz;
  cn_raw_dir(g, j, i, dx, dy, dz);
  cn_finish_ray(g.t[0], g.t[1], g.t[2], dx, dy, dz, g.vd, g.ndc, g.ax, g.ay, o, d, v);
}

// host side: the public descriptor -> the device one (validates it)
static inline int cn_make_raygen(const cnerf_raygen* c, RayGenDev* g) {
  if (!c || c->H <= 0 || c->W <= 0 || c->first < 0 || !(c->fx != 0.f) || !(c->fy != 0.f)) return CNERF_E_ARG;
  g->on = 1; g->W = c->W;
  g->fx = c->fx; g->fy = c->fy; g->cx = c->cx; g->cy = c->cy;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) g->r[3 * r + k] = c->c2w[4 * r + k];
    g->t[r] = c->c2w[4 * r + 3];
  }
  g->near = c->near; g->far = c->far; g->vd = c->use_viewdirs; g->ndc = c->ndc; g->ax = c->ndc_ax; g->ay = c->ndc_ay;
  g->first = c->first;
  return CNERF_OK;
}
static inline RayGenDev cn_no_raygen() {
  RayGenDev g = {};
  return g;
}
