// cnerf_render_fwd / cnerf_render_bwd: render_rays (R:311-421, V:441-551) and its autograd as ONE call each — the
// launch sequence the Python surface drives (run_nerf.py::render_rays), for callers that bind the C ABI directly:
//   coarse_z -> [gamma + MLP](coarse) -> composite -> (Nf > 0:) resample -> [gamma + MLP](fine) -> composite
// and backwards  composite_bwd -> dgrad -> wgrad  per level (no gradient through the resampling, R:397).
// Host code only: every launch goes through the public entry points of this library, on the caller's stream, inside
// the caller's workspace; nothing is allocated and nothing synchronises.
#include "raygen.hpp"

int cn_coarse_z_cam(float near, float far, int64_t B, int Nc, const float* t_vals, const float* t_rand, int lindisp,
                    float* z, hipStream_t st);
int cn_composite_fwd_cam(const float* raw, int raw_ch, const float* z, const RayGenDev& cam, const float* noise, int64_t B,
                         int S, int white_bkgd, float* rgb, float* disp, float* acc, float* depth, float* weights,
                         hipStream_t st);
int cn_mlp_fwd_cam(const cnerf_net* net, const float* packed, const RayGenDev& cam, const float* z, int64_t B, int S,
                   float* raw, void* stream);

namespace {

struct Layout {            // float offsets into the workspace
  int64_t z0, raw0, w0;    // coarse level: z[B,Nc], raw[B,Nc,C0], weights[B,Nc]
  int64_t z1, raw1, w1;    // fine level (Nf > 0): z[B,S1], raw[B,S1,C1], weights[B,S1]
  int64_t stash0, stash1;  // training stashes
  int64_t d_raw, bwd;      // backward scratch: d_raw of the coarse level, cnerf_mlp_bwd workspace of the coarse level
  int64_t d_raw1, bwd1;    // ... of the fine level (both levels are in flight together: cnerf_mlp_bwd_pair)
  int64_t total;
  int C0, C1, S1;
};

int raw_channels(const cnerf_net* n) { return n->use_viewdirs ? 4 : n->output_ch; }

int make_layout(const cnerf_net* coarse, const cnerf_net* fine, const cnerf_render_cfg* cfg, int64_t B, Layout* L) {
  if (!coarse || !cfg || B < 0 || cfg->Nc <= 0 || cfg->Nf < 0 || (cfg->ray_stride != 8 && cfg->ray_stride != 11))
    return CNERF_E_ARG;
  const cnerf_net* n1 = fine ? fine : coarse;      // R:402: the coarse network serves both levels when there is no fine one
  if ((coarse->use_viewdirs || n1->use_viewdirs) && cfg->ray_stride != 11) return CNERF_E_ARG;
  const int64_t Nc = cfg->Nc, S1 = cfg->Nc + cfg->Nf;
  L->C0 = raw_channels(coarse); L->C1 = raw_channels(n1); L->S1 = (int)S1;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += cn_round_up(n, 64); return at; };
  L->z0 = take(B * Nc); L->raw0 = take(B * Nc * L->C0); L->w0 = take(B * Nc);
  L->z1 = L->raw1 = L->w1 = L->stash0 = L->stash1 = -1;
  if (cfg->Nf > 0) { L->z1 = take(B * S1); L->raw1 = take(B * S1 * L->C1); L->w1 = take(B * S1); }
  L->d_raw = L->bwd = L->d_raw1 = L->bwd1 = -1;
  if (cfg->train) {
    L->stash0 = take(cnerf_mlp_stash_floats(coarse, B * Nc));
    if (cfg->Nf > 0) L->stash1 = take(cnerf_mlp_stash_floats(n1, B * S1));
    L->d_raw = take(B * Nc * L->C0);
    L->bwd = take(cnerf_mlp_bwd_ws_floats(coarse, B * Nc));
    if (cfg->Nf > 0) {
      L->d_raw1 = take(B * S1 * L->C1);
      L->bwd1 = take(cnerf_mlp_bwd_ws_floats(n1, B * S1));
    }
  }
  L->total = o;
  return CNERF_OK;
}

}  // namespace

extern "C" int64_t cnerf_render_ws_floats(const cnerf_net* coarse, const cnerf_net* fine, const cnerf_render_cfg* cfg,
                                          int64_t B) {
  Layout L;
  return make_layout(coarse, fine, cfg, B, &L) == CNERF_OK ? L.total : -1;
}

extern "C" int cnerf_render_fwd(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                                const float* packed_fine, const float* rays, int64_t B, const cnerf_render_cfg* cfg,
                                const float* t_vals, const float* t_rand, const float* u, int64_t u_row_stride,
                                const float* noise0, const float* noise1, const cnerf_render_out* out,
                                float* workspace, void* stream) {
  Layout L;
  int rc = make_layout(coarse, fine, cfg, B, &L);
  if (rc) return rc;
  if (!packed_coarse || !rays || !t_vals || !out || !workspace || (fine && !packed_fine) || (cfg->Nf > 0 && !u))
    return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  float* ws = workspace;
  const int Nc = cfg->Nc, rs = cfg->ray_stride;
  const bool two = cfg->Nf > 0;
  // coarse level (R:355-393)
  if ((rc = cnerf_coarse_z(rays, rs, B, Nc, t_vals, t_rand, cfg->lindisp, ws + L.z0, stream))) return rc;
  if ((rc = cnerf_mlp_fwd(coarse, packed_coarse, nullptr, rays, rs, nullptr, ws + L.z0, B, Nc, ws + L.raw0,
                          cfg->train ? ws + L.stash0 : nullptr, stream)))
    return rc;
  if ((rc = cnerf_composite_fwd(ws + L.raw0, L.C0, ws + L.z0, rays, rs, noise0, B, Nc, cfg->white_bkgd,
                                two ? out->rgb0 : out->rgb_map, two ? out->disp0 : out->disp_map,
                                two ? out->acc0 : out->acc_map, two ? out->depth0 : out->depth_map, ws + L.w0, stream)))
    return rc;
  const float *z_last = ws + L.z0, *raw_last = ws + L.raw0, *w_last = ws + L.w0;
  int S_last = Nc, C_last = L.C0;
  if (two) {   // R:395-404
    const cnerf_net* n1 = fine ? fine : coarse;
    const float* p1 = fine ? packed_fine : packed_coarse;
    if ((rc = cnerf_resample(ws + L.z0, ws + L.w0, u, u_row_stride, B, Nc, cfg->Nf, ws + L.z1, out->z_std, nullptr,
                             nullptr, stream)))
      return rc;
    if ((rc = cnerf_mlp_fwd(n1, p1, nullptr, rays, rs, nullptr, ws + L.z1, B, L.S1, ws + L.raw1,
                            cfg->train ? ws + L.stash1 : nullptr, stream)))
      return rc;
    if ((rc = cnerf_composite_fwd(ws + L.raw1, L.C1, ws + L.z1, rays, rs, noise1, B, L.S1, cfg->white_bkgd, out->rgb_map,
                                  out->disp_map, out->acc_map, out->depth_map, ws + L.w1, stream)))
      return rc;
    z_last = ws + L.z1; raw_last = ws + L.raw1; w_last = ws + L.w1; S_last = L.S1; C_last = L.C1;
  }
  // optional copies of the last level's per-sample tensors (retraw, R:412)
  hipStream_t st = cn_stream(stream);
  if (out->raw && hipMemcpyAsync(out->raw, raw_last, (size_t)B * S_last * C_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (out->z_vals && hipMemcpyAsync(out->z_vals, z_last, (size_t)B * S_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (out->weights && hipMemcpyAsync(out->weights, w_last, (size_t)B * S_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  return CNERF_OK;
}

extern "C" int cnerf_render_fwd_cam(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                                    const float* packed_fine, const cnerf_raygen* cam_in, int64_t B,
                                    const cnerf_render_cfg* cfg_in, const float* t_vals, const float* t_rand, const float* u,
                                    int64_t u_row_stride, const float* noise0, const float* noise1,
                                    const cnerf_render_out* out, float* workspace, void* stream) {
  if (!cfg_in || cfg_in->train) return CNERF_E_ARG;
  cnerf_render_cfg cfg = *cfg_in;
  cfg.ray_stride = 11;
  RayGenDev cam;
  int rc = cn_make_raygen(cam_in, &cam);
  if (rc) return rc;
  if (cam.first + B > (int64_t)cam_in->H * cam_in->W) return CNERF_E_ARG;
  Layout L;
  if ((rc = make_layout(coarse, fine, &cfg, B, &L))) return rc;
  if (!packed_coarse || !t_vals || !out || !workspace || (fine && !packed_fine) || (cfg.Nf > 0 && !u)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  float* ws = workspace;
  hipStream_t st = cn_stream(stream);
  const int Nc = cfg.Nc;
  const bool two = cfg.Nf > 0;
  if ((rc = cn_coarse_z_cam(cam.near, cam.far, B, Nc, t_vals, t_rand, cfg.lindisp, ws + L.z0, st))) return rc;
  if ((rc = cn_mlp_fwd_cam(coarse, packed_coarse, cam, ws + L.z0, B, Nc, ws + L.raw0, stream))) return rc;
  if ((rc = cn_composite_fwd_cam(ws + L.raw0, L.C0, ws + L.z0, cam, noise0, B, Nc, cfg.white_bkgd,
                                 two ? out->rgb0 : out->rgb_map, two ? out->disp0 : out->disp_map,
                                 two ? out->acc0 : out->acc_map, two ? out->depth0 : out->depth_map, ws + L.w0, st)))
    return rc;
  const float *z_last = ws + L.z0, *raw_last = ws + L.raw0, *w_last = ws + L.w0;
  int S_last = Nc, C_last = L.C0;
  if (two) {
    const cnerf_net* n1 = fine ? fine : coarse;
    const float* p1 = fine ? packed_fine : packed_coarse;
    if ((rc = cnerf_resample(ws + L.z0, ws + L.w0, u, u_row_stride, B, Nc, cfg.Nf, ws + L.z1, out->z_std, nullptr,
                             nullptr, stream)))
      return rc;
    if ((rc = cn_mlp_fwd_cam(n1, p1, cam, ws + L.z1, B, L.S1, ws + L.raw1, stream))) return rc;
    if ((rc = cn_composite_fwd_cam(ws + L.raw1, L.C1, ws + L.z1, cam, noise1, B, L.S1, cfg.white_bkgd, out->rgb_map,
                                   out->disp_map, out->acc_map, out->depth_map, ws + L.w1, st)))
      return rc;
    z_last = ws + L.z1; raw_last = ws + L.raw1; w_last = ws + L.w1; S_last = L.S1; C_last = L.C1;
  }
  if (out->raw && hipMemcpyAsync(out->raw, raw_last, (size_t)B * S_last * C_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (out->z_vals && hipMemcpyAsync(out->z_vals, z_last, (size_t)B * S_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (out->weights && hipMemcpyAsync(out->weights, w_last, (size_t)B * S_last * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  return CNERF_OK;
}

extern "C" int cnerf_render_bwd(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                                const float* packed_fine, const float* rays, int64_t B, const cnerf_render_cfg* cfg,
                                const float* noise0, const float* noise1, const cnerf_render_grads* g, float* workspace,
                                const cnerf_ptrs* grads_coarse, const cnerf_ptrs* grads_fine, int accumulate,
                                void* stream) {
  Layout L;
  int rc = make_layout(coarse, fine, cfg, B, &L);
  if (rc) return rc;
  if (!cfg->train || !packed_coarse || !rays || !g || !workspace || !grads_coarse || (fine && (!packed_fine || !grads_fine)))
    return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  float* ws = workspace;
  const int Nc = cfg->Nc, rs = cfg->ray_stride;
  const bool two = cfg->Nf > 0;
  if (two && fine) {
    // two networks: their backward passes are independent (the fine depths are detached, R:397) -> both compositing
    // backwards, then ONE dgrad grid + ONE wgrad grid for both (fine level first: the larger blocks lead the grid)
    if ((rc = cnerf_composite_bwd(ws + L.raw1, L.C1, ws + L.z1, rays, rs, noise1, B, L.S1, cfg->white_bkgd, g->g_rgb_map,
                                  g->g_disp_map, g->g_acc_map, g->g_depth_map, ws + L.d_raw1, stream)))
      return rc;
    if ((rc = cnerf_composite_bwd(ws + L.raw0, L.C0, ws + L.z0, rays, rs, noise0, B, Nc, cfg->white_bkgd, g->g_rgb0,
                                  g->g_disp0, g->g_acc0, g->g_depth0, ws + L.d_raw, stream)))
      return rc;
    return cnerf_mlp_bwd_pair(fine, packed_fine, ws + L.d_raw1, B, L.S1, ws + L.stash1, ws + L.bwd1, grads_fine,
                              coarse, packed_coarse, ws + L.d_raw, B, Nc, ws + L.stash0, ws + L.bwd, grads_coarse,
                              accumulate, stream);
  }
  int acc0 = accumulate;
  if (two) {   // one network serving both levels (R:402): fine level first (the order autograd runs it)
    if ((rc = cnerf_composite_bwd(ws + L.raw1, L.C1, ws + L.z1, rays, rs, noise1, B, L.S1, cfg->white_bkgd, g->g_rgb_map,
                                  g->g_disp_map, g->g_acc_map, g->g_depth_map, ws + L.d_raw1, stream)))
      return rc;
    if ((rc = cnerf_mlp_bwd(coarse, packed_coarse, ws + L.d_raw1, B, L.S1, ws + L.stash1, ws + L.bwd1, grads_coarse,
                            accumulate, stream)))
      return rc;
    acc0 = 1;   // the coarse pass adds to what the fine pass wrote
  }
  if ((rc = cnerf_composite_bwd(ws + L.raw0, L.C0, ws + L.z0, rays, rs, noise0, B, Nc, cfg->white_bkgd,
                                two ? g->g_rgb0 : g->g_rgb_map, two ? g->g_disp0 : g->g_disp_map,
                                two ? g->g_acc0 : g->g_acc_map, two ? g->g_depth0 : g->g_depth_map, ws + L.d_raw, stream)))
    return rc;
  return cnerf_mlp_bwd(coarse, packed_coarse, ws + L.d_raw, B, Nc, ws + L.stash0, ws + L.bwd, grads_coarse, acc0, stream);
}
