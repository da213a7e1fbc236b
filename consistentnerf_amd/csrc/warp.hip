// Cross-view depth warp and hard-mask precompute — the ConsistentNeRF contribution:
// get_ref_rays / get_test_label (V:576-669) and the mask loop of train() (V:994-1046).
// One thread per world point / target pixel; per-5120-pixel chunk threshold search is one workgroup
// with an LDS min-reduction (replaces a host-synchronising `while mask.sum()==0` loop per chunk).
#include "raygen.hpp"

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


// ---- a15: the in-loop consistency block's ray construction as ONE launch (VT:905-925 + get_ref_rays VT:451-501) ---------------
// For the N rays of the batch: P = rays_o + depth * rays_d (VT:905), projection into the reference camera (`project`, no axis flip in
// the VT variant), strict bounds -> mask_bound; the in-bounds points, COMPACTED in batch order (what the reference's boolean
// indexing `x[mask]` does with a host sync each), define rays of the reference camera through the snapped pixels: direction
// ((px - cx) / fx, (py - cy) / fy, 1) rotated by c2w (get_rays_ref V:553-574), origin = the camera centre; colour / depth prior
// of the reference view at the pixel; |z_cam - D_ref| and from its minimum the occlusion threshold thr0 * 2^k (smallest k that
// lets one point pass: the reference's `while mask.sum() == 0` doubling, VT:921-925, a host sync per iteration there).
// One workgroup of 1024 threads walks the batch in chunks (ballot + popcount scan: the compaction keeps batch order and is
// deterministic); a second sweep applies the threshold.  Also written: the [M, 8|11] rows render() would assemble from the new
// rays (raygen.hpp: the arithmetic of pack_rays_k) and sel[N] = mask_bound AND occlusion mask per PRIMARY ray (what
// `x[mask_bound][mask]` selects, VT:941-969) as a 0/1 weight.
struct SsDev {
  Mat34 w2c;
  RayGenDev cam;            // the reference camera + the bounds / flags of the render() the rays are for
  int H, W, flip, image_ch;
  float thr0;
};

// cnerf_ss_batch: the same launch also assembles the ONE batch the whole `--ss_loss` step renders (run_nerf_view.ss_step_loss):
// rows [0, N) = the batch's own rays packed as render() packs them, rows [N, N + M) = the reference rays (the `rows` output of the
// kernel, pointed there by the host), rows [N + M, 2 N) = padding (the reference camera's optical axis: a valid ray whose loss
// weight is 0), and next to it the per-row target colour / depth prior / loss mask and the LIVE row count N + M — all in device
// memory: the host never learns M (no read-back; the MLP launches take the count from `live`).
struct SsComb {
  float* rows;              // [2N, 8|11]  or nullptr (= plain cnerf_ss_ref_rays)
  float* target;            // [2N, 3]: target_s | reference colours at the snapped pixels | 0
  float* prior;             // [2N]:    depth (the batch's prior) | reference depth prior there | 0
  float* mask;              // [2N]:    sel | 1 | 0
  int32_t* live;            // [1]:     N + M
  const float* target_s;    // [N, 3]
  const float* amin_in;     // the GLOBAL minimum of |z - D_ref| over a batch sharded across ranks (device, 1 float), or nullptr
};

__global__ __launch_bounds__(1024) void ss_ref_rays_k(SsDev a, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                      const float* __restrict__ depth, int64_t N, const float* __restrict__ image,
                                                      const float* __restrict__ depth_ref, float* __restrict__ rows,
                                                      float* __restrict__ rays_od, float* __restrict__ target,
                                                      float* __restrict__ depth_tgt, float* __restrict__ depth_diff,
                                                      uint8_t* __restrict__ inb, uint8_t* __restrict__ mask,
                                                      float* __restrict__ sel, int32_t* __restrict__ rank, int32_t* __restrict__ meta,
                                                      SsComb cb) {
  __shared__ int wcnt[16];
  __shared__ float wmin[16];
  __shared__ int wnan[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int base = 0;                      // in-bounds points in front of this chunk (workgroup-uniform)
  float amin = __builtin_inff();
  int anynan = 0;
  for (int64_t c0 = 0; c0 < N; c0 += 1024) {
    const int64_t i = c0 + tid;
    Proj p = {};
    bool in = false;
    if (i < N) {
      const float dep = depth[i];
      p = project(rays_o[3 * i] + dep * rays_d[3 * i], rays_o[3 * i + 1] + dep * rays_d[3 * i + 1],
                  rays_o[3 * i + 2] + dep * rays_d[3 * i + 2], a.w2c, a.cam.fx, a.cam.fy, a.cam.cx, a.cam.cy, a.H, a.W, a.flip);
      in = p.inb;
      if (cb.rows) {          // the primary ray's own row, colour and prior (the arithmetic of pack_rays_k: raygen.hpp)
        const int rs = a.cam.vd ? 11 : 8;
        float o[3], d[3], v[3];
        cn_finish_ray(rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2], rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2], a.cam.vd,
                      a.cam.ndc, a.cam.ax, a.cam.ay, o, d, v);
        float* out = cb.rows + i * rs;
        out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
        out[6] = a.cam.near; out[7] = a.cam.far;
        if (a.cam.vd) { out[8] = v[0]; out[9] = v[1]; out[10] = v[2]; }
        cb.target[3 * i] = cb.target_s[3 * i]; cb.target[3 * i + 1] = cb.target_s[3 * i + 1]; cb.target[3 * i + 2] = cb.target_s[3 * i + 2];
        cb.prior[i] = dep;
      }
    }
    const unsigned long long bal = __ballot(in);
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base, tot = 0;
    for (int w = 0; w < 16; ++w) {
      off += w < wv ? wcnt[w] : 0;
      tot += wcnt[w];
    }
    if (in) {
      const int64_t j = off + __popcll(bal & ((1ull << lane) - 1ull));
      const float d0 = (p.px - a.cam.cx) / a.cam.fx, d1 = (p.py - a.cam.cy) / a.cam.fy, d2 = 1.f;   // VT:491
      const float dx = d0 * a.cam.r[0] + d1 * a.cam.r[1] + d2 * a.cam.r[2];                        // directions @ c2w[:3,:3].T (V:566)
      const float dy = d0 * a.cam.r[3] + d1 * a.cam.r[4] + d2 * a.cam.r[5];
      const float dz = d0 * a.cam.r[6] + d1 * a.cam.r[7] + d2 * a.cam.r[8];
      if (rays_od) {
        float* ro = rays_od + j * 3;
        float* rd = rays_od + (N + j) * 3;
        ro[0] = a.cam.t[0]; ro[1] = a.cam.t[1]; ro[2] = a.cam.t[2];
        rd[0] = dx; rd[1] = dy; rd[2] = dz;
      }
      if (rows) {
        float o[3], d[3], v[3];
        cn_finish_ray(a.cam.t[0], a.cam.t[1], a.cam.t[2], dx, dy, dz, a.cam.vd, a.cam.ndc, a.cam.ax, a.cam.ay, o, d, v);
        float* out = rows + j * (a.cam.vd ? 11 : 8);
        out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
        out[6] = a.cam.near; out[7] = a.cam.far;
        if (a.cam.vd) { out[8] = v[0]; out[9] = v[1]; out[10] = v[2]; }
      }
      const int64_t pix = (int64_t)(int)p.py * a.W + (int)p.px;
      if (target) {
        const float* q = image + pix * a.image_ch;
        target[3 * j] = q[0]; target[3 * j + 1] = q[1]; target[3 * j + 2] = q[2];
      }
      const float dr = depth_ref[pix];
      if (depth_tgt) depth_tgt[j] = dr;
      const float ad = fabsf(p.zc - dr);
      depth_diff[j] = ad;
      if (ad != ad) anynan = 1;
      amin = ad < amin ? ad : amin;
      rank[i] = (int32_t)j;
    } else if (i < N) {
      rank[i] = -1;
    }
    if (i < N && inb) inb[i] = in ? 1 : 0;
    base += tot;
    __syncthreads();   // wcnt is rewritten by the next chunk
  }
  // minimum of |diff| over the in-bounds points (torch.min: NaN wins)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(amin, o, 64);
    amin = t < amin ? t : amin;
    anynan |= __shfl_xor(anynan, o, 64);
  }
  if (lane == 0) { wmin[wv] = amin; wnan[wv] = anynan; }
  __syncthreads();   // (also: every depth_diff / rank store of this workgroup is visible below)
  amin = wmin[0];
  anynan = wnan[0];
  for (int w = 1; w < 16; ++w) {
    amin = wmin[w] < amin ? wmin[w] : amin;
    anynan |= wnan[w];
  }
  const int M = base;
  // thr = thr0 * 2^k, smallest k >= 0 with min|diff| < thr (VT:921-925 doubles until something passes).  A NaN |diff| (a NaN depth
  // prior) passes no threshold — `NaN < thr` is false, exactly as in the reference's loop — so the minimum is taken over the other
  // points (`amin` above never picks a NaN up) and the doubling goes on until one of THOSE passes (it used to stop at
  // k = 0 whenever a NaN was present).  Only when no point has a finite |diff| at all does k reach the cap (the reference loops
  // forever there): meta[3] reports NaNs, the mask is then empty.
  const float amin_local = amin;
  if (cb.amin_in) amin = cb.amin_in[0];     // a batch sharded over ranks: every rank applies the threshold of the WHOLE batch
  float thr = a.thr0;
  int k = 0;
  while (!(amin < thr) && k < 63) { thr = 2.f * thr; ++k; }
  if (tid == 0) {
    meta[0] = M; meta[1] = k; meta[2] = __float_as_int(thr); meta[3] = anynan;
    meta[4] = __float_as_int(amin_local);
    if (cb.live) cb.live[0] = (int32_t)(N + M);
  }
  if (mask)
    for (int64_t j = tid; j < M; j += 1024) mask[j] = depth_diff[j] < thr ? 1 : 0;
  int nsel = 0;
  for (int64_t i = tid; i < N; i += 1024) {
    const int32_t j = rank[i];
    const float sv = (j >= 0 && depth_diff[j] < thr) ? 1.f : 0.f;
    nsel += sv != 0.f;
    if (sel) sel[i] = sv;
    if (cb.mask) cb.mask[i] = sv;
  }
  // meta[5] = how many primary rays `sel` selects (with M and N: the three counts a sharded step all-reduces)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nsel += __shfl_xor(nsel, o, 64);
  __syncthreads();                   // (wcnt is free: the chunk loop is over)
  if (lane == 0) wcnt[wv] = nsel;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wcnt[w];
    meta[5] = t;
  }
  if (cb.rows) {
    // second segment: live rows [N, N + M) weigh 1, the padding rows [N + M, 2 N) are a valid ray (the reference camera's optical
    // axis) with zero target / prior and weight 0 — nothing downstream has to know where the live rows end to stay finite
    const int rs = a.cam.vd ? 11 : 8;
    float o[3], d[3], v[3];
    cn_finish_ray(a.cam.t[0], a.cam.t[1], a.cam.t[2], a.cam.r[2], a.cam.r[5], a.cam.r[8], a.cam.vd, a.cam.ndc, a.cam.ax, a.cam.ay, o, d, v);
    for (int64_t j = tid; j < N; j += 1024) {
      cb.mask[N + j] = j < M ? 1.f : 0.f;
      if (j >= M) {
        float* out = cb.rows + (N + j) * rs;
        out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
        out[6] = a.cam.near; out[7] = a.cam.far;
        if (a.cam.vd) { out[8] = v[0]; out[9] = v[1]; out[10] = v[2]; }
        cb.target[3 * (N + j)] = 0.f; cb.target[3 * (N + j) + 1] = 0.f; cb.target[3 * (N + j) + 2] = 0.f;
        cb.prior[N + j] = 0.f;
      }
    }
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

extern "C" int cnerf_ss_ref_rays(const cnerf_ss_warp* c, const float* rays_o, const float* rays_d, const float* depth, int64_t N,
                                 const float* image, const float* depth_ref, float* rows, float* rays_od, float* target,
                                 float* depth_tgt, float* depth_diff, uint8_t* inb, uint8_t* mask, float* sel, int32_t* rank,
                                 int32_t* meta, void* stream) {
  if (!c || !rays_o || !rays_d || !depth || !depth_ref || !depth_diff || !rank || !meta || N <= 0 || N >= (1ll << 31) ||
      c->ref.H < 2 || c->ref.W < 2 || !(c->thr0 > 0.f) || (target && (!image || c->image_ch < 3)))
    return CNERF_E_ARG;
  SsDev a = {};
  cnerf_raygen rg = c->ref;
  rg.first = 0;
  int rc = cn_make_raygen(&rg, &a.cam);
  if (rc) return rc;
  a.w2c = load34(c->w2c);
  a.H = c->ref.H; a.W = c->ref.W; a.flip = c->flip; a.image_ch = c->image_ch; a.thr0 = c->thr0;
  hipLaunchKernelGGL(ss_ref_rays_k, dim3(1), dim3(1024), 0, cn_stream(stream), a, rays_o, rays_d, depth, N, image, depth_ref, rows,
                     rays_od, target, depth_tgt, depth_diff, inb, mask, sel, rank, meta, SsComb{});
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_ss_batch(const cnerf_ss_warp* c, const float* rays_o, const float* rays_d, const float* depth, const float* target_s,
                              int64_t N, const float* image, const float* depth_ref, const float* amin_global, float* rows2,
                              float* target2, float* prior2, float* mask2, int32_t* live, float* rays_od, float* depth_diff, uint8_t* inb,
                              uint8_t* mask, float* sel, int32_t* rank, int32_t* meta, void* stream) {
  if (!c || !rays_o || !rays_d || !depth || !target_s || !depth_ref || !image || !rows2 || !target2 || !prior2 || !mask2 || !live ||
      !depth_diff || !rank || !meta || N <= 0 || N >= (1ll << 30) || c->ref.H < 2 || c->ref.W < 2 || !(c->thr0 > 0.f) || c->image_ch < 3)
    return CNERF_E_ARG;
  SsDev a = {};
  cnerf_raygen rg = c->ref;
  rg.first = 0;
  int rc = cn_make_raygen(&rg, &a.cam);
  if (rc) return rc;
  a.w2c = load34(c->w2c);
  a.H = c->ref.H; a.W = c->ref.W; a.flip = c->flip; a.image_ch = c->image_ch; a.thr0 = c->thr0;
  const int rs = a.cam.vd ? 11 : 8;
  SsComb cb{rows2, target2, prior2, mask2, live, target_s, amin_global};
  hipLaunchKernelGGL(ss_ref_rays_k, dim3(1), dim3(1024), 0, cn_stream(stream), a, rays_o, rays_d, depth, N, image, depth_ref,
                     rows2 + N * rs, rays_od, target2 + 3 * N, prior2 + N, depth_diff, inb, mask, sel, rank, meta, cb);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
