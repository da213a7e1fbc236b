// Cross-view depth warp and hard-mask precompute — the ConsistentNeRF contribution:
// get_ref_rays / get_test_label (V:576-669) and the mask loop of train() (V:994-1046).
// One thread per world point / target pixel; per-5120-pixel chunk threshold search is one workgroup
// with an LDS min-reduction (replaces a host-synchronising `while mask.sum()==0` loop per chunk).
#include "common.hpp"

namespace {

struct Mat34 { float r[9]; float t[3]; };

static Mat34 load34(const float* h) {
  Mat34 m;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) m.r[3 * r + k] = h[4 * r + k];
    m.t[r] = h[4 * r + 3];
  }
  return m;
}

struct Proj {
  float xc, yc, zc;  // camera-frame point (after the optional OpenGL->OpenCV flip)
  float px, py;      // rounded pixel (half-to-even), as floats
  bool inb;
};

// (P R^T + T) [diag(1,-1,-1)] -> K -> round -> strict bounds  (V:592-613)
__device__ __forceinline__ Proj project(float X, float Y, float Z, const Mat34& w2c, float fx, float fy, float cx,
                                        float cy, int H, int W, int flip) {
  Proj p;
  p.xc = X * w2c.r[0] + Y * w2c.r[1] + Z * w2c.r[2] + w2c.t[0];
  p.yc = X * w2c.r[3] + Y * w2c.r[4] + Z * w2c.r[5] + w2c.t[1];
  p.zc = X * w2c.r[6] + Y * w2c.r[7] + Z * w2c.r[8] + w2c.t[2];
  if (flip) { p.yc = -p.yc; p.zc = -p.zc; }
  const float ux = p.xc * fx + p.zc * cx;   // intrinsics are [[fx,0,cx],[0,fy,cy],[0,0,1]]
  const float uy = p.yc * fy + p.zc * cy;
  p.px = rintf(ux / p.zc + 0.0f);
  p.py = rintf(uy / p.zc + 0.0f);
  const float xn = p.px / (float)(W - 1), yn = p.py / (float)(H - 1);
  p.inb = (xn > 0.f) && (xn < 1.f) && (yn > 0.f) && (yn < 1.f);
  return p;
}

__global__ void warp_points_k(const float* __restrict__ P, int64_t N, Mat34 w2c, float fx, float fy, float cx,
                              float cy, int H, int W, int flip, float* __restrict__ Xc, float* __restrict__ px,
                              float* __restrict__ py, uint8_t* __restrict__ inb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Proj p = project(P[3 * i], P[3 * i + 1], P[3 * i + 2], w2c, fx, fy, cx, cy, H, W, flip);
  if (Xc) { Xc[3 * i] = p.xc; Xc[3 * i + 1] = p.yc; Xc[3 * i + 2] = p.zc; }
  if (px) px[i] = p.px;
  if (py) py[i] = p.py;
  if (inb) inb[i] = p.inb ? 1 : 0;
}

constexpr int KMAX = 300;   // thr0 * 2^k overflows to +inf long before this: every finite |diff| passes

// smallest k >= 0 with diff < thr0 * 2^k (thr doubled in fp32 exactly like V:1026-1029)
__device__ __forceinline__ int pass_level(float diff, float thr0) {
  float thr = thr0;
  int k = 0;
  while (!(diff < thr) && k < KMAX) { thr = 2.f * thr; ++k; }
  return k;
}

__global__ __launch_bounds__(256) void hard_mask_k(int H, int W, float fx, float fy, float cx, float cy, Mat34 c2w_t,
                                                   Mat34 w2c_r, const float* __restrict__ depth_t,
                                                   const float* __restrict__ depth_r, float thr0, int chunk,
                                                   uint8_t* __restrict__ mask, float* __restrict__ thr_out) {
  __shared__ int kmin;
  if (threadIdx.x == 0) kmin = KMAX + 1;
  __syncthreads();
  const int64_t npix = (int64_t)H * W;
  const int64_t base = (int64_t)blockIdx.x * chunk;
  auto level = [&](int64_t idx) -> int {
    const int j = (int)(idx / W), i = (int)(idx - (int64_t)j * W);
    // target ray through pixel (i, j): get_rays H:164-173, then P = o + depth * d (V:1015-1016)
    const float d0 = ((float)i - cx) / fx, d1 = -((float)j - cy) / fy, d2 = -1.f;
    const float dx = d0 * c2w_t.r[0] + d1 * c2w_t.r[1] + d2 * c2w_t.r[2];
    const float dy = d0 * c2w_t.r[3] + d1 * c2w_t.r[4] + d2 * c2w_t.r[5];
    const float dz = d0 * c2w_t.r[6] + d1 * c2w_t.r[7] + d2 * c2w_t.r[8];
    const float dep = depth_t[idx];
    const Proj p = project(c2w_t.t[0] + dep * dx, c2w_t.t[1] + dep * dy, c2w_t.t[2] + dep * dz, w2c_r, fx, fy, cx,
                           cy, H, W, 1);
    if (!p.inb) return KMAX + 1;
    const float dr = depth_r[(int64_t)(int)p.py * W + (int)p.px];
    return pass_level(fabsf(p.zc - dr), thr0);
  };
  int local = KMAX + 1;
  for (int o = threadIdx.x; o < chunk; o += blockDim.x) {
    const int64_t idx = base + o;
    if (idx < npix) {
      const int k = level(idx);
      local = k < local ? k : local;
    }
  }
  atomicMin(&kmin, local);
  __syncthreads();
  const int km = kmin;
  if (threadIdx.x == 0 && thr_out) {
    float thr = thr0;
    for (int k = 0; k < km && k < KMAX; ++k) thr = 2.f * thr;
    thr_out[blockIdx.x] = km > KMAX ? __builtin_nanf("") : thr;
  }
  if (km > KMAX) return;   // no in-bounds pixel in this chunk: mask untouched (V:1037-1038)
  for (int o = threadIdx.x; o < chunk; o += blockDim.x) {
    const int64_t idx = base + o;
    if (idx < npix && level(idx) == km) mask[idx] = 1;   // OR over reference views (V:1041)
  }
}

}  // namespace

extern "C" int cnerf_warp_points(const float* P, int64_t N, const float* w2c_host, float fx, float fy, float cx,
                                 float cy, int H, int W, int flip, float* Xc, float* px, float* py, uint8_t* inb,
                                 void* stream) {
  if (!P || !w2c_host || N < 0 || H < 2 || W < 2) return CNERF_E_ARG;
  if (N == 0) return CNERF_OK;
  hipLaunchKernelGGL(warp_points_k, dim3((unsigned)cn_div_up(N, 256)), dim3(256), 0, cn_stream(stream), P, N,
                     load34(w2c_host), fx, fy, cx, cy, H, W, flip, Xc, px, py, inb);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_hard_mask_pair(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_tgt_host,
                                    const float* w2c_ref_host, const float* depth_tgt, const float* depth_ref,
                                    float thr0, int chunk, uint8_t* mask, float* thr_out, void* stream) {
  if (!c2w_tgt_host || !w2c_ref_host || !depth_tgt || !depth_ref || !mask || H < 2 || W < 2 || chunk <= 0 ||
      !(thr0 > 0.f))
    return CNERF_E_ARG;
  const int64_t npix = (int64_t)H * W;
  hipLaunchKernelGGL(hard_mask_k, dim3((unsigned)cn_div_up(npix, chunk)), dim3(256), 0, cn_stream(stream), H, W, fx,
                     fy, cx, cy, load34(c2w_tgt_host), load34(w2c_ref_host), depth_tgt, depth_ref, thr0, chunk, mask,
                     thr_out);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
