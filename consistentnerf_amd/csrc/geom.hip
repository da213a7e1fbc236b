// Network geometry, parameter-tensor bookkeeping and the weight re-pack kernel.
// (Replaces nothing in the reference: it is the bridge between the nn.Parameter layout the reference
//  checkpoints use — H:86-101 — and the MFMA operand panels described in common.hpp.)
#include <math.h>
#include <string.h>

#include "common.hpp"

static int enc_dim(int L) { return L < 0 ? 3 : 3 + 6 * L; }

int cn_make_geom(const cnerf_net* net, NetGeom* g) {
  if (!net || !g) return CNERF_E_ARG;
  memset(g, 0, sizeof(*g));
  if (net->W != 64 && net->W != 128 && net->W != 256) return CNERF_E_UNSUPPORTED;
  if (net->D < 1 || net->D > 16) return CNERF_E_UNSUPPORTED;
  if (net->multires > 10 || net->multires < -1) return CNERF_E_UNSUPPORTED;
  if (net->use_viewdirs && (net->multires_views > 4 || net->multires_views < -1)) return CNERF_E_UNSUPPORTED;
  if (net->skip >= 0 && net->D == net->skip + 1) return CNERF_E_UNSUPPORTED;  // reference itself mis-shapes here
  if (!net->use_viewdirs && (net->output_ch < 4 || net->output_ch > 8)) return CNERF_E_UNSUPPORTED;
  g->D = net->D; g->W = net->W; g->NT = net->W / 32; g->Wh = net->W / 2;
  g->L = net->multires; g->Ld = net->multires_views;
  g->in_ch = enc_dim(net->multires); g->in_chp = (int)cn_round_up(g->in_ch, 32);
  g->viewdirs = net->use_viewdirs ? 1 : 0;
  g->dir_ch = g->viewdirs ? enc_dim(net->multires_views) : 0;
  g->dir_chp = (int)cn_round_up(g->dir_ch, 32);
  g->out_ch = g->viewdirs ? 4 : net->output_ch;
  g->skip = (net->skip >= 0 && net->skip + 1 < net->D) ? net->skip : -1;
  const int64_t W = g->W, Wh = g->Wh;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += cn_round_up(n, 64); return o; };
  g->f_l0 = take((int64_t)(g->in_chp + 8) * W);            // + bias group (common.hpp)
  for (int l = 1; l < g->D; ++l) g->f_trunk[l] = take((W + 8) * W);
  g->f_skip = g->skip >= 0 ? take((int64_t)(g->in_chp + 8) * W) : -1;
  for (int l = 1; l < g->D; ++l) g->t_trunk[l] = take(W * W);
  for (int l = 0; l < g->D; ++l) g->b_trunk[l] = take(W);
  if (g->viewdirs) {
    g->f_feat = take((W + 8) * W);
    g->f_views = take(W * Wh);
    g->f_viewsd = take((int64_t)(g->dir_chp + 8) * Wh);
    g->t_feat = take(W * W);
    g->t_views = take(Wh * W);
    g->v_alpha = take(W);
    g->v_rgb = take(3 * Wh);
    g->b_feat = take(W); g->b_views = take(Wh); g->b_alpha = take(1); g->b_rgb = take(3);
    g->v_out = g->b_out = -1;
  } else {
    g->v_out = take((int64_t)g->out_ch * W);
    g->b_out = take(g->out_ch);
  }
  g->total = off;
  int r = 0;
  g->s_enc = r; r += g->in_chp;
  for (int l = 0; l < g->D; ++l) { g->s_h[l] = r; r += g->W; }
  if (g->viewdirs) {
    g->s_feat = r; r += g->W;
    g->s_denc = r; r += g->dir_chp;
    g->s_hv = r; r += g->Wh;
  }
  {
    const int md = (g->NT + 1) / 2, mdv = (g->NT / 2 + 1) / 2;   // dwords per half-wave: trunk layer / view branch
    int o = 0;
    for (int l = 0; l < g->D; ++l) { g->s_mb[l] = o; o += 2 * md; }
    g->s_mb[g->D] = o;
    if (g->viewdirs) o += 2 * mdv;
    g->s_mask = r; r += (int)cn_round_up(o, 8);   // every block starts on a column octet (tile-major storage)
  }
  g->s_rows = r;
  r = 0;
  for (int l = 0; l < g->D; ++l) { g->g_z[l] = r; r += g->W; }
  if (g->viewdirs) {
    g->g_feat = r; r += g->W;
    g->g_hv = r; r += g->Wh;
  }
  g->g_out = r; r += 32;
  g->g_rows = r;
  return CNERF_OK;
}

extern "C" int cnerf_num_tensors(const cnerf_net* net) {
  if (!net) return CNERF_E_ARG;
  return 2 * net->D + (net->use_viewdirs ? 8 : 4);
}

extern "C" int cnerf_tensor_shape(const cnerf_net* net, int i, int64_t* rows, int64_t* cols) {
  NetGeom g;
  int rc = cn_make_geom(net, &g);
  if (rc) return rc;
  const int n = cnerf_num_tensors(net);
  if (i < 0 || i >= n || !rows || !cols) return CNERF_E_ARG;
  int64_t r = 0, c = 0;
  const int D = g.D;
  // NB: the reference builds the skip layer whenever 4 is in range(D-1) (H:86-87)
  const bool has_skip_layer = net->skip >= 0 && net->skip + 1 < D;
  if (i < 2 * D) {
    const int l = i / 2;
    r = g.W;
    c = l == 0 ? g.in_ch : (has_skip_layer && l == net->skip + 1 ? g.W + g.in_ch : g.W);
  } else {
    switch (i - 2 * D) {
      case 0: case 1: r = g.Wh; c = g.W + g.dir_ch; break;
      case 2: case 3: if (g.viewdirs) { r = g.W; c = g.W; } else { r = g.out_ch; c = g.W; } break;
      case 4: case 5: r = 1; c = g.W; break;
      case 6: case 7: r = 3; c = g.Wh; break;
    }
  }
  if (i & 1) c = 1;
  *rows = r; *cols = c;
  return CNERF_OK;
}

extern "C" int64_t cnerf_packed_floats(const cnerf_net* net) {
  NetGeom g;
  if (cn_make_geom(net, &g)) return -1;
  return g.total;
}

extern "C" int64_t cnerf_mlp_stash_floats(const cnerf_net* net, int64_t M) {
  NetGeom g;
  if (cn_make_geom(net, &g) || M < 0) return -1;
  return (int64_t)g.s_rows * cn_round_up(M, 32);
}

extern "C" int cnerf_abi_version(void) { return CNERF_ABI_VERSION; }

extern "C" const char* cnerf_strerror(int code) {
  switch (code) {
    case CNERF_OK: return "ok";
    case CNERF_E_ARG: return "invalid argument";
    case CNERF_E_UNSUPPORTED: return "configuration outside the compiled envelope";
    case CNERF_E_NODEVICE: return "no gfx950 device";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int cnerf_device_info(int dev, char* name64, int* num_cus, int* lds_bytes) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return CNERF_E_NODEVICE;
  if (name64) { strncpy(name64, p.gcnArchName, 63); name64[63] = 0; }
  if (num_cus) *num_cus = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? CNERF_OK : CNERF_E_NODEVICE;
}

// ------------------------------------------------------------------------------------------------
// pack kernel: a table of jobs, one grid.y slice each
namespace {

enum { JOB_PANEL = 0, JOB_PANEL_T = 1, JOB_COPY = 2 };
struct PackJob {
  const float* src;   // weight [N, ld] (or vector)
  int ld, col0, N, K; // source rows/cols used
  int mode;
  int rows_p;         // padded row count of the panel (N rounded to 32 for PANEL; K rounded to 32 for PANEL_T)
  int groups;         // number of 8-wide groups along the contracted index
  float* out;         // where the panel goes (the packed buffer + the panel's offset)
  const float* bias;  // JOB_PANEL only: extra group `groups` with P[groups][n][0] = bias[n] (the bias MFMA step)
  int zero_bias;      // JOB_PANEL with bias == nullptr: still write the extra group, as zeros
};
constexpr int MAX_JOBS = 56;
struct PackArgs { PackJob job[MAX_JOBS]; };   // (56 x 64 B: inside the 4 KiB kernarg segment)

// (the job table is read in place from the kernarg segment: taken by value and indexed with blockIdx.y the compiler copies the
// whole 3.6 KB struct to scratch in every thread first — measured 26 us per network instead of ~6)
__global__ void pack_k(PackArgs a_by_value) {
  (void)a_by_value;
  const __attribute__((address_space(4))) PackArgs& a = *(const __attribute__((address_space(4))) PackArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const __attribute__((address_space(4))) PackJob& j = a.job[blockIdx.y];
  const int ngroups = j.groups + (j.mode == JOB_PANEL && (j.bias != nullptr || j.zero_bias) ? 1 : 0);
  const int64_t n = j.mode == JOB_COPY ? (int64_t)j.N * j.K : (int64_t)ngroups * j.rows_p * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (j.mode == JOB_COPY) {
      const int r = (int)(i / j.K), c = (int)(i % j.K);
      v = j.src[(int64_t)r * j.ld + j.col0 + c];
    } else {
      const int e = (int)(i & 7);
      const int row = (int)((i >> 3) % j.rows_p);
      const int grp = (int)((i >> 3) / j.rows_p);
      if (j.mode == JOB_PANEL) {          // P[kg][n][e] = W[n][col0 + 8kg + e]
        const int k = 8 * grp + e;
        if (grp == j.groups) { if (e == 0 && row < j.N && j.bias != nullptr) v = j.bias[row]; }
        else if (row < j.N && k < j.K) v = j.src[(int64_t)row * j.ld + j.col0 + k];
      } else {                            // PT[ng][k][e] = W[8ng + e][col0 + k]
        const int nn = 8 * grp + e;
        if (nn < j.N && row < j.K) v = j.src[(int64_t)nn * j.ld + j.col0 + row];
      }
    }
    j.out[i] = v;
  }
}

}  // namespace

// Appends one network's panels to the job table (a.job[nj...]); CNERF_E_UNSUPPORTED when the table is full.
static int add_pack_jobs(const cnerf_net* net, const cnerf_ptrs* params, float* packed, PackArgs& a, int& nj) {
  NetGeom g;
  int rc = cn_make_geom(net, &g);
  if (rc) return rc;
  if (!params || !packed) return CNERF_E_ARG;
  const int nt = cnerf_num_tensors(net);
  for (int i = 0; i < nt; ++i)
    if (!params->p[i] && !(i >= 2 * g.D && i < 2 * g.D + 2 && !g.viewdirs)) return CNERF_E_ARG;
  bool full = false;
  const int W = g.W, Wh = g.Wh, D = g.D;
  // contracted width padded to a multiple of 32 (4 K-groups of 8: the GEMM pipelines peel / unroll by that)
  auto panel = [&](const float* src, int ld, int col0, int N, int K, int64_t dst, const float* bias = nullptr,
                   int zero_bias = 0) {
    if (nj >= MAX_JOBS) { full = true; return; }
    a.job[nj++] = PackJob{src, ld, col0, N, K, JOB_PANEL, (int)cn_round_up(N, 32), (int)cn_round_up(K, 32) / 8, packed + dst,
                          bias, zero_bias};
  };
  auto panel_t = [&](const float* src, int ld, int col0, int N, int K, int64_t dst) {
    if (nj >= MAX_JOBS) { full = true; return; }
    a.job[nj++] = PackJob{src, ld, col0, N, K, JOB_PANEL_T, (int)cn_round_up(K, 32), (int)cn_div_up(N, 8), packed + dst,
                          nullptr, 0};
  };
  auto copy = [&](const float* src, int N, int K, int64_t dst) {
    if (nj >= MAX_JOBS) { full = true; return; }
    a.job[nj++] = PackJob{src, K, 0, N, K, JOB_COPY, 0, 0, packed + dst, nullptr, 0};
  };
  auto Wt = [&](int l) { return params->p[2 * l]; };
  auto Bt = [&](int l) { return params->p[2 * l + 1]; };
  panel(Wt(0), g.in_ch, 0, W, g.in_ch, g.f_l0, Bt(0));
  for (int l = 1; l < D; ++l) {
    const bool sk = g.skip >= 0 && l == g.skip + 1;
    const int ld = sk ? W + g.in_ch : W, c0 = sk ? g.in_ch : 0;
    panel(Wt(l), ld, c0, W, W, g.f_trunk[l], sk ? nullptr : Bt(l), 1);   // skip layer: bias rides on f_skip (zeros here)
    panel_t(Wt(l), ld, c0, W, W, g.t_trunk[l]);
    if (sk) panel(Wt(l), ld, 0, W, g.in_ch, g.f_skip, Bt(l));
  }
  const int base = 2 * D;
  if (g.viewdirs) {
    const float* Wv = params->p[base + 0];
    const int ldv = W + g.dir_ch;
    panel(params->p[base + 2], W, 0, W, W, g.f_feat, params->p[base + 3]);
    panel_t(params->p[base + 2], W, 0, W, W, g.t_feat);
    panel(Wv, ldv, 0, Wh, W, g.f_views);
    panel(Wv, ldv, W, Wh, g.dir_ch, g.f_viewsd, params->p[base + 1]);
    panel_t(Wv, ldv, 0, Wh, W, g.t_views);
    copy(params->p[base + 4], 1, W, g.v_alpha);
    copy(params->p[base + 6], 3, Wh, g.v_rgb);
    copy(params->p[base + 5], 1, 1, g.b_alpha);
    copy(params->p[base + 7], 1, 3, g.b_rgb);
  } else {
    copy(params->p[base + 2], g.out_ch, W, g.v_out);
    copy(params->p[base + 3], 1, g.out_ch, g.b_out);
  }
  return full ? CNERF_E_UNSUPPORTED : CNERF_OK;
}

extern "C" int cnerf_pack_weights(const cnerf_net* net, const cnerf_ptrs* params, float* packed, void* stream) {
  PackArgs a;
  int nj = 0;
  const int rc = add_pack_jobs(net, params, packed, a, nj);
  if (rc) return rc;
  hipLaunchKernelGGL(pack_k, dim3(64, nj), dim3(256), 0, cn_stream(stream), a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// Both networks of a render_rays call in ONE launch (the coarse and the fine network are re-packed together after every optimizer
// step); two launches when their panels do not fit one job table.
extern "C" int cnerf_pack_weights_pair(const cnerf_net* net0, const cnerf_ptrs* params0, float* packed0, const cnerf_net* net1,
                                       const cnerf_ptrs* params1, float* packed1, void* stream) {
  PackArgs a;
  int nj = 0;
  int rc = add_pack_jobs(net0, params0, packed0, a, nj);
  if (rc) return rc;
  const int n0 = nj;
  rc = add_pack_jobs(net1, params1, packed1, a, nj);
  if (rc == CNERF_E_UNSUPPORTED) {      // table full: the first network now, the second on its own
    hipLaunchKernelGGL(pack_k, dim3(64, n0), dim3(256), 0, cn_stream(stream), a);
    CN_CHECK_LAUNCH();
    return cnerf_pack_weights(net1, params1, packed1, stream);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(pack_k, dim3(64, nj), dim3(256), 0, cn_stream(stream), a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
