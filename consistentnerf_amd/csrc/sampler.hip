// The training batch of ONE image as one launch: the pixel choice of run_nerf_view.train() V:1452-1517 (the
// pixels of P patches first, row index fastest inside a patch V:1490-1494, then N_rand DISTINCT pixels of the — optionally
// centre-cropped — grid, `np.random.choice(..., replace=False)` V:1503) and of run_nerf.train() R:730-757 (P = 0), the rays of
// those pixels (get_rays H:164-173) in both forms the callers want — raw (rays_o, rays_d) [2, B, 3] like the reference's
// `batch_rays`, and the [B, 8|11] rows render() assembles from them (R:100-125: NDC warp, near / far, view directions) — the
// colours target_s [B, 3] and up to 4 per-pixel maps (depth prior, hard mask, monocular depth) gathered at the same pixels.
// Replaces, per step: cnerf_gen_rays over the WHOLE image (8.4 MB at 378 x 504), torch.randperm over the grid (8 merge-sort + 1
// radix-sort launches), 6 ATen index kernels, 2 cat, 2 arange, a stack and cnerf_pack_rays.
//
// The distinct pixels: when the caller does not pass its own draw (`select_inds`), pixel k is pi(k) for a keyed pseudo-random
// PERMUTATION pi of [0, n) — any prefix of a permutation is a draw without replacement.  pi = an alternating (unbalanced) Feistel
// network on ceil(log2 n) bits, 8 rounds, round function = Philox4x32-10 of (round, half) under the stream's (seed, offset),
// cycle-walked into [0, n) (x -> pi2(x) until < n: < 2 expected applications).  Each round xors one half with a function of the
// other, so every round — and the walk — is a bijection.  oracle/philox.py::permutation restates it in numpy; tests compare bit
// for bit, check distinctness / range / uniformity, and that `select_inds` reproduces the reference's own batch (fixture `patch`).
#include "raygen.hpp"
#include "rng.hpp"

namespace {

struct PixDev {
  RayGenDev cam;
  int crop_r0, crop_c0, crop_w;
  uint32_t n_grid;
  int bits_a, bits_b;
  int n_patch_rays, ps;
  int start[16][2];
  int image_ch, n_extras;
  int H;
  const float* extras[4];
};

__device__ __forceinline__ uint32_t perm_index(uint32_t i, uint32_t n, int a, int b, uint64_t key, uint64_t off) {
  if (n <= 1) return 0;
  const uint32_t ma = (1u << a) - 1u, mb = (1u << b) - 1u;
  uint32_t x = i;
  do {
    uint32_t L = x >> b, R = x & mb;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      L ^= cn_philox_x0((uint64_t)R | ((uint64_t)r << 32), off, key) & ma;
      R ^= cn_philox_x0((uint64_t)L | ((uint64_t)(r + 1) << 32), off, key) & mb;
    }
    x = (L << b) | R;
  } while (x >= n);
  return x;
}

__global__ void sample_pixels_k(PixDev p, int64_t B, const int64_t* __restrict__ select, CnRngK rk,
                                const float* __restrict__ image, float* __restrict__ rays, float* __restrict__ rays_od,
                                float* __restrict__ target, float* __restrict__ extras_out, int64_t* __restrict__ coords) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int row, col;
  if (b < p.n_patch_rays) {
    const int n = p.ps * p.ps, q = (int)(b / n), k = (int)(b - (int64_t)q * n);
    row = p.start[q][0] + k % p.ps;          // V:1490-1494: the row index runs fastest inside a patch
    col = p.start[q][1] + k / p.ps;
  } else {
    const int64_t k = b - p.n_patch_rays;
    uint32_t g;
    if (select) {
      // caller-supplied indices address the image and the priors: clamp (a negative or too large index must not become an
      // out-of-bounds read; raybank._sample range-checks them and raises / asserts — ADVICE r05)
      const int64_t sv = select[k];
      g = sv < 0 ? 0u : (sv >= (int64_t)p.n_grid ? (uint32_t)(p.n_grid - 1) : (uint32_t)sv);
    } else {
      const uint64_t seed = rk.dev ? rk.dev[0] : rk.seed;
      g = perm_index((uint32_t)k, p.n_grid, p.bits_a, p.bits_b, seed ^ 0x636e6572665f7078ull /* "cnerf_px" */,
                     rk.offset + (rk.dev ? rk.dev[1] : 0ull));
    }
    row = p.crop_r0 + (int)(g / (uint32_t)p.crop_w);
    col = p.crop_c0 + (int)(g % (uint32_t)p.crop_w);
  }
  float dx, dy, dz, o[3], d[3], v[3];
  cn_raw_dir(p.cam, row, col, dx, dy, dz);
  if (rays_od) {                              // the reference's batch_rays = stack([rays_o, rays_d]) (R:755, V:1515)
    float* ro = rays_od + b * 3;
    float* rd = rays_od + (B + b) * 3;
    ro[0] = p.cam.t[0]; ro[1] = p.cam.t[1]; ro[2] = p.cam.t[2];
    rd[0] = dx; rd[1] = dy; rd[2] = dz;
  }
  cn_finish_ray(p.cam.t[0], p.cam.t[1], p.cam.t[2], dx, dy, dz, p.cam.vd, p.cam.ndc, p.cam.ax, p.cam.ay, o, d, v);
  if (rays) {
    float* out = rays + b * (p.cam.vd ? 11 : 8);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
    out[6] = p.cam.near; out[7] = p.cam.far;
    if (p.cam.vd) { out[8] = v[0]; out[9] = v[1]; out[10] = v[2]; }
  }
  const int64_t pix = (int64_t)row * p.cam.W + col;
  if (target) {
    const float* px = image + pix * p.image_ch;
    target[b * 3 + 0] = px[0]; target[b * 3 + 1] = px[1]; target[b * 3 + 2] = px[2];
  }
  for (int e = 0; e < p.n_extras; ++e) extras_out[(int64_t)e * B + b] = p.extras[e][pix];
  if (coords) { coords[2 * b] = row; coords[2 * b + 1] = col; }
}

}  // namespace

extern "C" int cnerf_sample_pixels(const cnerf_pixel_batch* c, const int64_t* select_inds, const cnerf_rng* rng, const float* image,
                                   const float* const* extras, float* rays, float* rays_od, float* target, float* extras_out,
                                   int64_t* coords, void* stream) {
  if (!c || c->H <= 0 || c->W <= 0 || c->n_rand < 0 || c->n_patches < 0 || c->n_patches > 16 || c->n_extras < 0 ||
      c->n_extras > 4 || (c->n_patches > 0 && c->patch_size <= 0) || (target && (!image || c->image_ch < 3)) ||
      (c->n_extras > 0 && (!extras || !extras_out)))
    return CNERF_E_ARG;
  if (c->crop_r0 < 0 || c->crop_c0 < 0 || c->crop_h < 0 || c->crop_w < 0 || c->crop_r0 + c->crop_h > c->H ||
      c->crop_c0 + c->crop_w > c->W)
    return CNERF_E_ARG;
  const int64_t n_grid = (int64_t)c->crop_h * c->crop_w;
  if (c->n_rand > 0 && (n_grid <= 0 || n_grid >= (1ll << 31))) return CNERF_E_ARG;
  if (c->n_rand > 0 && !select_inds && (!rng || c->n_rand > n_grid)) return CNERF_E_ARG;   // a draw WITHOUT replacement
  PixDev p = {};
  cnerf_raygen rg;
  rg.H = c->H; rg.W = c->W; rg.fx = c->fx; rg.fy = c->fy; rg.cx = c->cx; rg.cy = c->cy;
  for (int k = 0; k < 12; ++k) rg.c2w[k] = c->c2w[k];
  rg.near = c->near; rg.far = c->far; rg.use_viewdirs = c->use_viewdirs; rg.ndc = c->ndc; rg.ndc_ax = c->ndc_ax; rg.ndc_ay = c->ndc_ay;
  rg.first = 0;
  int rc = cn_make_raygen(&rg, &p.cam);
  if (rc) return rc;
  p.H = c->H;
  p.crop_r0 = c->crop_r0; p.crop_c0 = c->crop_c0; p.crop_w = c->crop_w > 0 ? c->crop_w : 1;
  p.n_grid = (uint32_t)n_grid;
  int bits = 2;
  while ((1ll << bits) < n_grid) ++bits;
  p.bits_a = bits / 2; p.bits_b = bits - p.bits_a;
  p.ps = c->patch_size > 0 ? c->patch_size : 1;
  p.n_patch_rays = c->n_patches * p.ps * p.ps;
  for (int q = 0; q < c->n_patches; ++q) {
    p.start[q][0] = c->patch_start[q][0]; p.start[q][1] = c->patch_start[q][1];
    if (p.start[q][0] < 0 || p.start[q][1] < 0 || p.start[q][0] + p.ps > c->H || p.start[q][1] + p.ps > c->W) return CNERF_E_ARG;
  }
  p.image_ch = c->image_ch; p.n_extras = c->n_extras;
  for (int e = 0; e < c->n_extras; ++e) {
    if (!extras[e]) return CNERF_E_ARG;
    p.extras[e] = extras[e];
  }
  const int64_t B = p.n_patch_rays + c->n_rand;
  if (B == 0) return CNERF_OK;
  CnRngK rk = {};
  if (rng) rk = cn_rng_arg(rng);
  hipLaunchKernelGGL(sample_pixels_k, dim3((unsigned)cn_div_up(B, 256)), dim3(256), 0, cn_stream(stream), p, B, select_inds, rk,
                     image, rays, rays_od, target, extras_out, coords);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
