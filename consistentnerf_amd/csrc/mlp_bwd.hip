// Backward of the fused encoding+MLP (autograd of run_network R:37-52 / NeRF.forward H:107-130).
// Three launches:
//   1. dgrad (this file): one wave64 per 32 points walks the network backwards with the TRANSPOSED weight
//      panels as the MFMA A operand and the previous gradient's accumulator registers as B (mlp_common.hpp: no
//      LDS, no barriers); it writes the gradient w.r.t. every layer's pre-activation, dZ_l, into the tile-major
//      gradient workspace G[Mp][g_rows].
//   2. wgrad (wgrad.hip): NT GEMMs contracted over points, dW_l = dZ_l^T . H_{l-1}, split over point ranges.
//   3. a fixed-order reduction of the split partials into the parameter gradients (deterministic).
// ReLU masks come from the sign-bit words the forward packed into the stash (1 bit per hidden unit, s_mask).
#include "mlp_common.hpp"
#include "timing.hpp"

int cn_wgrad_launch(const NetGeom& g, const float* stash, const float* G, int64_t M, int64_t Mp, float* partials,
                    int nsplit, const cnerf_ptrs* grads, int accumulate, hipStream_t st, int bf3 = 0);
int cn_wgrad_launch_n(int n, const NetGeom* const* g, const float* const* stash, const float* const* G, const int64_t* Mp,
                      float* const* partials, const int* nsplit, const cnerf_ptrs* const* grads, int accumulate,
                      hipStream_t st, int bf3 = 0, const int* live = nullptr, const int* live_mul = nullptr,
                      const int* live_sub = nullptr);
int cn_wgrad_nsplit(int64_t Mp);
int64_t cn_param_floats(const NetGeom& g);

namespace {

// One level's operands.  A launch carries up to two levels of the SAME architecture (the coarse and the fine network of a
// training step: independent once the forward is done): blocks [0, nb0) walk level 0, the rest level 1 — one grid, so the
// 8 rounds of the coarse level ride behind the 24 of the fine one instead of paying their own ramp and tail.
struct BwdLevel {
  const float* packed;
  const float* d_raw;
  const float* stash;
  float* G;
  int64_t M, Mp;
  int64_t live_mul;   // points per ray of this level (with BwdArgs::live), 0 = no gating
  int64_t live_sub;   // rays in front of this level's arrays that are not part of the launch (first_ray of the _live calls)
};

struct BwdArgs {
  NetGeom g;
  BwdLevel lv[2];
  unsigned nb0;
  const int* live;    // device count of LIVE rays or nullptr: tiles at or beyond live * live_mul points retire at once (their raw
                      // outputs were zeros, their gradient tile rows are never read: wgrad clips its point ranges the same way)
};

#define CN_CONST __attribute__((address_space(4)))

template <int NT, bool VD>
__global__ __launch_bounds__(64) void mlp_dgrad_k(BwdArgs args_by_value) {
  constexpr int W = NT * 32;
  constexpr int NTH = NT / 2 > 0 ? NT / 2 : 1;
  constexpr int MD = (NT + 1) / 2, MDV = (NTH + 1) / 2;
  (void)args_by_value;   // read in place from the kernarg segment (scalar loads; the level is picked by blockIdx.x)
  const CN_CONST BwdArgs& args = *(const CN_CONST BwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const CN_CONST NetGeom& g = args.g;
  const unsigned nb0 = args.nb0;
  const bool second = blockIdx.x >= nb0;
  const CN_CONST BwdLevel& a = args.lv[second ? 1 : 0];
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p0 = (int64_t)(blockIdx.x - (second ? nb0 : 0u)) * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (int)(a.M - p0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
  if (args.live != nullptr && a.live_mul > 0 && p0 >= ((int64_t)args.live[0] - a.live_sub) * a.live_mul) return;   // padding rays
  CN_TINIT(1)
  const APanel AP{make_rsrc(a.packed, (unsigned)(g.total * 4)), (m * 8 + 4 * hh) * 4};
  // this workgroup's stash tile row (sign bits) and gradient tile row (tile-major, mlp_common.hpp); lanes of padding
  // points address out of range: their bits read as 0 and their stores are dropped (the launcher zero-fills the last
  // tile row of G for the wgrad DMA)
  const rsrc_t srs = make_rsrc(a.stash + p0 * g.s_rows, (unsigned)(32 * g.s_rows * 4));
  const rsrc_t grs = make_rsrc(a.G + p0 * g.g_rows, (unsigned)(32 * g.g_rows * 4));
  const bool valid = p < a.M;
  const int gvo = valid ? m * 32 + hh * 16 : TM_OOB;
  const int smo = valid ? m * 32 + hh * MD * 4 : TM_OOB;
  f32x16 X[NT], Y[NT];
  f32x4 A[3][NT];   // A-operand register sets of the current transposed panel
  unsigned bits[MD];

  if (VD) {
    const float4 d = *reinterpret_cast<const float4*>(a.d_raw + pc * 4);
    const float dc[4] = {d.x, d.y, d.z, d.w};
    unsigned bv[MDV];
    load_bits<MDV>(srs, valid ? m * 32 + hh * MDV * 4 : TM_OOB, tm_col(g.s_mask + g.s_mb[g.D]), bv);
    a_prefetch3<NT>(A, AP, (int)g.t_views, W, g.Wh / 8 - 1);
    if (hh == 0) buf_store(grs, valid ? m * 32 : TM_OOB, tm_col(g.g_out), f32x4{d.x, d.y, d.z, d.w});
    // rgb_linear^T on the VALU, masked by the view-branch ReLU -> dZv (C-layout registers)
    f32x16 V[NTH];
    {
      f32x4 wq[3][NTH][4];   // all weight quads in flight before the first use: one exposed L2 round trip, not 12*NTH
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < NTH; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            wq[c][t][q] = buf_load(AP.rs, hh * 16, (int)(g.v_rgb + (int64_t)c * g.Wh + 32 * t + 8 * q) * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NTH; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) sacc += wq[c][t][q][j] * dc[c];
            V[t][4 * q + j] = sacc;
          }
    }
    mask_bits<NTH>(V, bv);
    CN_T(0)
    // dF = views_linears^T (feature columns only; gamma(d) needs no gradient) . dZv, no mask (feature_linear is linear)
    gemm_reg3<NTH, NT, false, true>(X, V, A, AP, (int)g.t_views, W, hh, TileStores<NTH, NT>{V, grs, gvo, tm_col(g.g_hv)});
    pin<NT>(X);
    CN_T(2)
    // dZ_{D-1} = relu'(h_{D-1}) * (feature_linear^T . dF + alpha_linear^T . dsigma)
    // (only the first A set before the sigma-head quads: all three next to X, the quads and their products do not fit the
    //  arch-VGPR half of the register file; sets 1 and 2 are needed 32 and 64 MFMAs into the GEMM)
    a_load<NT>(A[0], AP, (int)g.t_feat, W, 0);
    load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[g.D - 1]), bits);
    {
      // the weight quads land in Y's own registers and are scaled in place (all loads in flight before the first use: one
      // exposed L2 round trip; a separate staging array would be 128 more live registers next to X, Y and A)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = buf_load(AP.rs, hh * 16, (int)(g.v_alpha + 32 * t + 8 * q) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) Y[t][4 * q + j] = wq[j];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[t][r] *= dc[3];
      a_load<NT>(A[1], AP, (int)g.t_feat, W, 1);
      a_load<NT>(A[2], AP, (int)g.t_feat, W, 2);
    }
    CN_T(4)
    gemm_reg3<NT, NT, false, false>(Y, X, A, AP, (int)g.t_feat, W, hh, TileStores<NT, NT>{X, grs, gvo, tm_col(g.g_feat)});
    CN_T(2)
  } else {
    load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[g.D - 1]), bits);
    float dc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dc[c] = c < g.out_ch ? a.d_raw[pc * g.out_ch + c] : 0.f;
    if (hh == 0)
      for (int c = 0; c < g.out_ch; ++c)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dc[c]), grs, valid ? m * 32 : TM_OOB,
                                              tm_col(g.g_out + c), 0);
    // output_linear^T on the VALU
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < g.out_ch) {
            const f32x4 w = buf_load(AP.rs, hh * 16, (int)(g.v_out + (int64_t)c * W + 32 * t + 8 * q) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += w[j] * dc[c];
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) Y[t][4 * q + j] = s[j];
      }
    CN_T(0)
  }
  if (g.D > 1) a_prefetch3<NT>(A, AP, (int)g.t_trunk[g.D - 1], W, W / 8 - 1);
  mask_bits<NT>(Y, bits);
  CN_T(3)
  // trunk: dZ_{l-1} = relu'(h_{l-1}) * (W_l^T . dZ_l) (the gamma(x) columns of the skip layer get no gradient); dZ_l
  // goes out to the workspace while it is the B operand of this GEMM.  X / Y alternate as input and output.
  auto layer = [&](f32x16 (&In)[NT], f32x16 (&Out)[NT], int l) __attribute__((always_inline)) {
    load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[l - 1]), bits);
    gemm_reg3<NT, NT, false, true>(Out, In, A, AP, (int)g.t_trunk[l], W, hh, TileStores<NT, NT>{In, grs, gvo, tm_col(g.g_z[l])});
    if (l > 1) a_prefetch3<NT>(A, AP, (int)g.t_trunk[l - 1], W, W / 8 - 1);
    CN_T(2)
    mask_bits<NT>(Out, bits);
    CN_T(3)
  };
  int l = g.D - 1;
  for (; l >= 2; l -= 2) {
    layer(Y, X, l);
    layer(X, Y, l - 1);
  }
  if (l == 1) {   // (a third instance of the layer body: cheaper than keeping both sets live behind a flag)
    layer(Y, X, 1);
    store_tiles<NT>(X, grs, gvo, tm_col(g.g_z[0]));
  } else {
    store_tiles<NT>(Y, grs, gvo, tm_col(g.g_z[0]));
  }
  CN_T(3)
  CN_TEND
}

template <int NT>
int launch(const BwdArgs& a, int nlev, hipStream_t st) {
  unsigned grid = 0;
  for (int i = 0; i < nlev; ++i) {
    const BwdLevel& L = a.lv[i];
    grid += (unsigned)cn_div_up(L.M, 32);
    if (L.Mp > L.M) {   // last gradient tile row holds padding points: the kernel drops their stores, wgrad reads them
      hipError_t e = hipMemsetAsync(L.G + (L.Mp - 32) * a.g.g_rows, 0, (size_t)32 * a.g.g_rows * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (a.g.viewdirs) hipLaunchKernelGGL((mlp_dgrad_k<NT, true>), dim3(grid), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((mlp_dgrad_k<NT, false>), dim3(grid), dim3(64), 0, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

int dispatch(const BwdArgs& a, int nlev, hipStream_t st) {
  switch (a.g.NT) {
    case 2: return launch<2>(a, nlev, st);
    case 4: return launch<4>(a, nlev, st);
    case 8: return launch<8>(a, nlev, st);
  }
  return CNERF_E_UNSUPPORTED;
}

}  // namespace

#ifdef CN_TIMING
CN_TIMING_ACCESSOR(cnerf_debug_timing_bwd)
#endif

extern "C" int64_t cnerf_mlp_bwd_ws_floats(const cnerf_net* net, int64_t M) {
  NetGeom g;
  if (cn_make_geom(net, &g) || M < 0) return -1;
  const int64_t Mp = cn_round_up(M, 32);
  return (int64_t)g.g_rows * Mp + (int64_t)cn_wgrad_nsplit(Mp) * cn_round_up(cn_param_floats(g), 64);
}

static int dgrad_one(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S, const float* stash,
                     float* workspace, const int32_t* live, void* stream, int64_t first = 0) {
  BwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !d_raw || !stash || !workspace || B < 0 || S <= 0 || (live && S % 32 != 0)) return CNERF_E_ARG;
  if (first && (!live || first < 0 || first >= B)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  const int64_t o = first * S;      // (first_ray: operands advanced past the rays that are left out, see dgrad_pair)
  d_raw += o * (a.g.viewdirs ? 4 : a.g.out_ch);
  stash += o * a.g.s_rows;
  B -= first;
  a.lv[0] = BwdLevel{packed, d_raw, stash, workspace, B * S, cn_round_up(B * S, 32), live ? S : 0, first};
  a.lv[1] = a.lv[0];
  a.nb0 = (unsigned)cn_div_up(B * S, 32);
  a.live = live;
  return dispatch(a, 1, cn_stream(stream));
}
extern "C" int cnerf_mlp_dgrad(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                               const float* stash, float* workspace, void* stream) {
  return dgrad_one(net, packed, d_raw, B, S, stash, workspace, nullptr, stream);
}

static int wgrad_one(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace, const cnerf_ptrs* grads,
                     int accumulate, void* stream, int bf3, const int32_t* live = nullptr);
extern "C" int cnerf_mlp_wgrad(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace,
                               const cnerf_ptrs* grads, int accumulate, void* stream) {
  return wgrad_one(net, B, S, stash, workspace, grads, accumulate, stream, 0);
}
// OPT-IN bf16x3 weight gradients (second bench line only): the wide GEMMs on the bf16 matrix cores at three planes per operand
extern "C" int cnerf_mlp_wgrad_bf(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace,
                                  const cnerf_ptrs* grads, int accumulate, void* stream) {
  return wgrad_one(net, B, S, stash, workspace, grads, accumulate, stream, 1);
}
static int wgrad_one(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace, const cnerf_ptrs* grads,
                     int accumulate, void* stream, int bf3, const int32_t* live) {
  NetGeom g;
  int rc = cn_make_geom(net, &g);
  if (rc) return rc;
  if (!stash || !workspace || !grads || B < 0 || S <= 0 || (live && S % 32 != 0)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  const int64_t M = B * S, Mp = cn_round_up(M, 32);
  const int nsplit = cn_wgrad_nsplit(Mp);
  float* partials = workspace + (int64_t)g.g_rows * Mp;
  if (live) {
    const NetGeom* gp = &g;
    return cn_wgrad_launch_n(1, &gp, &stash, &workspace, &Mp, &partials, &nsplit, &grads, accumulate, cn_stream(stream), bf3, live, &S);
  }
  return cn_wgrad_launch(g, stash, workspace, M, Mp, partials, nsplit, grads, accumulate, cn_stream(stream), bf3);
}

extern "C" int cnerf_mlp_bwd(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                             const float* stash, float* workspace, const cnerf_ptrs* grads, int accumulate,
                             void* stream) {
  if (!grads) return CNERF_E_ARG;
  int rc = cnerf_mlp_dgrad(net, packed, d_raw, B, S, stash, workspace, stream);
  if (rc) return rc;
  return cnerf_mlp_wgrad(net, B, S, stash, workspace, grads, accumulate, stream);
}

// cnerf_mlp_bwd of a batch padded to a fixed capacity of B rays whose LIVE row count sits in device memory (cnerf_mlp_fwd_live)
extern "C" int cnerf_mlp_bwd_live(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                                  const float* stash, float* workspace, const cnerf_ptrs* grads, int accumulate,
                                  const int32_t* live_rays, void* stream) {
  if (!grads || !live_rays) return CNERF_E_ARG;
  int rc = dgrad_one(net, packed, d_raw, B, S, stash, workspace, live_rays, stream);
  if (rc) return rc;
  return wgrad_one(net, B, S, stash, workspace, grads, accumulate, stream, 0, live_rays);
}

// Backward of TWO independent networks in one dgrad grid + one wgrad grid (+ one reduction): the coarse and the fine
// network of a render_rays training step (R:311-421) — their backward passes share nothing once the forward is done
// (the fine level's sample depths are detached, R:397).  The dgrad grid is shared when both have the same architecture,
// otherwise two dgrad launches; the wgrad grid is always shared.  net0 / net1 must be different parameter sets (two
// reductions into one gradient tensor would race).  Workspaces as for cnerf_mlp_bwd, one per network.
static int dgrad_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0, const float* stash0,
                      float* workspace0, const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                      const float* stash1, float* workspace1, const int32_t* live, void* stream, int64_t first0 = 0, int64_t first1 = 0);
extern "C" int cnerf_mlp_dgrad_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                                    const float* stash0, float* workspace0, const cnerf_net* net1, const float* packed1,
                                    const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1,
                                    void* stream) {
  return dgrad_pair(net0, packed0, d_raw0, B0, S0, stash0, workspace0, net1, packed1, d_raw1, B1, S1, stash1, workspace1, nullptr,
                    stream);
}
static int dgrad_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0, const float* stash0,
                      float* workspace0, const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                      const float* stash1, float* workspace1, const int32_t* live, void* stream, int64_t first0, int64_t first1) {
  BwdArgs a;
  NetGeom g1;
  int rc = cn_make_geom(net0, &a.g);
  if (rc) return rc;
  if ((rc = cn_make_geom(net1, &g1))) return rc;
  if (!packed0 || !d_raw0 || !stash0 || !workspace0 || !packed1 || !d_raw1 || !stash1 || !workspace1 || B0 < 0 || B1 < 0 ||
      S0 <= 0 || S1 <= 0)
    return CNERF_E_ARG;
  const int64_t M0 = B0 * S0, M1 = B1 * S1;
  const bool same = net0->D == net1->D && net0->W == net1->W && net0->multires == net1->multires &&
                    net0->multires_views == net1->multires_views && net0->use_viewdirs == net1->use_viewdirs &&
                    net0->output_ch == net1->output_ch && net0->skip == net1->skip;
  if (live && (S0 % 32 != 0 || S1 % 32 != 0)) return CNERF_E_ARG;
  if ((first0 || first1) && (!live || first0 < 0 || first1 < 0 || first0 >= B0 || first1 >= B1)) return CNERF_E_ARG;
  if (same && M0 > 0 && M1 > 0) {
    // first_ray: the level's first `first` rays carry zero seeds and are left out — operands advanced past them (tile rows are
    // 32 points: first * S is a multiple of 32), the device-side count reduced by the same
    const int64_t o0 = first0 * S0, o1 = first1 * S1;
    const int rc0 = a.g.viewdirs ? 4 : a.g.out_ch;      // floats per point of d_raw (same architecture: same for both levels)
    a.lv[0] = BwdLevel{packed0, d_raw0 + o0 * rc0, stash0 + o0 * a.g.s_rows, workspace0, M0 - o0, cn_round_up(M0 - o0, 32), live ? S0 : 0, first0};
    a.lv[1] = BwdLevel{packed1, d_raw1 + o1 * rc0, stash1 + o1 * g1.s_rows, workspace1, M1 - o1, cn_round_up(M1 - o1, 32), live ? S1 : 0, first1};
    a.nb0 = (unsigned)cn_div_up(M0 - o0, 32);
    a.live = live;
    return dispatch(a, 2, cn_stream(stream));
  }
  // (two architectures: one dgrad launch per network)
  if ((rc = dgrad_one(net0, packed0, d_raw0, B0, S0, stash0, workspace0, live, stream, first0))) return rc;
  return dgrad_one(net1, packed1, d_raw1, B1, S1, stash1, workspace1, live, stream, first1);
}

static int wgrad_two(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                     const cnerf_net* net1, int64_t B1, int S1, const float* stash1, float* workspace1,
                     const cnerf_ptrs* grads1, int accumulate, void* stream, int bf3, const int32_t* live = nullptr, int64_t first0 = 0,
                     int64_t first1 = 0);
extern "C" int cnerf_mlp_wgrad_pair(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                                    const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1,
                                    const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate,
                                    void* stream) {
  return wgrad_two(net0, B0, S0, stash0, workspace0, grads0, net1, B1, S1, stash1, workspace1, grads1, accumulate, stream, 0);
}
extern "C" int cnerf_mlp_wgrad_bf_pair(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                                       const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1,
                                       const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate,
                                       void* stream) {
  return wgrad_two(net0, B0, S0, stash0, workspace0, grads0, net1, B1, S1, stash1, workspace1, grads1, accumulate, stream, 1);
}
static int wgrad_two(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                     const cnerf_net* net1, int64_t B1, int S1, const float* stash1, float* workspace1,
                     const cnerf_ptrs* grads1, int accumulate, void* stream, int bf3, const int32_t* live, int64_t first0,
                     int64_t first1) {
  if (!grads0 || !grads1 || !stash0 || !stash1 || !workspace0 || !workspace1 || B0 < 0 || B1 < 0 || S0 <= 0 || S1 <= 0 ||
      (live && (S0 % 32 != 0 || S1 % 32 != 0)))
    return CNERF_E_ARG;
  if ((first0 || first1) && (!live || first0 < 0 || first1 < 0 || first0 >= B0 || first1 >= B1)) return CNERF_E_ARG;
  if (B0 == 0) return wgrad_one(net1, B1, S1, stash1, workspace1, grads1, accumulate, stream, bf3, live);
  if (B1 == 0) return wgrad_one(net0, B0, S0, stash0, workspace0, grads0, accumulate, stream, bf3, live);
  for (int i = 0; i < CNERF_MAX_TENSORS; ++i)
    if (grads0->p[i] && grads0->p[i] == grads1->p[i]) return CNERF_E_ARG;
  NetGeom g0, g1;
  int rc = cn_make_geom(net0, &g0);
  if (rc) return rc;
  if ((rc = cn_make_geom(net1, &g1))) return rc;
  const int64_t Mp0 = cn_round_up((B0 - first0) * S0, 32), Mp1 = cn_round_up((B1 - first1) * S1, 32);
  stash0 += first0 * S0 * g0.s_rows;      // (first_ray: see dgrad_pair; the gradient workspace holds only the launched rays' rows)
  stash1 += first1 * S1 * g1.s_rows;
  const NetGeom* gs[2] = {&g0, &g1};
  const float* stashes[2] = {stash0, stash1};
  const float* Gs[2] = {workspace0, workspace1};
  const int64_t Mps[2] = {Mp0, Mp1};
  const int ns[2] = {cn_wgrad_nsplit(Mp0), cn_wgrad_nsplit(Mp1)};
  float* parts[2] = {workspace0 + (int64_t)g0.g_rows * Mp0, workspace1 + (int64_t)g1.g_rows * Mp1};
  const cnerf_ptrs* grs[2] = {grads0, grads1};
  const int muls[2] = {S0, S1};
  const int subs[2] = {(int)first0, (int)first1};
  return cn_wgrad_launch_n(2, gs, stashes, Gs, Mps, parts, ns, grs, accumulate, cn_stream(stream), bf3, live, live ? muls : nullptr,
                           live ? subs : nullptr);
}

extern "C" int cnerf_mlp_bwd_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                                  const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                                  const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                                  const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate,
                                  void* stream) {
  if (!grads0 || !grads1) return CNERF_E_ARG;
  for (int i = 0; i < CNERF_MAX_TENSORS; ++i)
    if (grads0->p[i] && grads0->p[i] == grads1->p[i]) return CNERF_E_ARG;
  int rc = cnerf_mlp_dgrad_pair(net0, packed0, d_raw0, B0, S0, stash0, workspace0, net1, packed1, d_raw1, B1, S1, stash1,
                                workspace1, stream);
  if (rc) return rc;
  return cnerf_mlp_wgrad_pair(net0, B0, S0, stash0, workspace0, grads0, net1, B1, S1, stash1, workspace1, grads1, accumulate,
                              stream);
}

// the two halves of cnerf_mlp_bwd_pair_live, separately launchable (cf. cnerf_mlp_dgrad_pair / cnerf_mlp_wgrad_pair)
extern "C" int cnerf_mlp_dgrad_pair_live(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                                         const float* stash0, float* workspace0, const cnerf_net* net1, const float* packed1,
                                         const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1,
                                         const int32_t* live_rays, int64_t first_ray0, int64_t first_ray1, void* stream) {
  if (!live_rays || B0 != B1) return CNERF_E_ARG;
  return dgrad_pair(net0, packed0, d_raw0, B0, S0, stash0, workspace0, net1, packed1, d_raw1, B1, S1, stash1, workspace1, live_rays,
                    stream, first_ray0, first_ray1);
}
extern "C" int cnerf_mlp_wgrad_pair_live(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                                         const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1, const float* stash1,
                                         float* workspace1, const cnerf_ptrs* grads1, int accumulate, const int32_t* live_rays,
                                         int64_t first_ray0, int64_t first_ray1, void* stream) {
  if (!live_rays || B0 != B1 || !grads0 || !grads1) return CNERF_E_ARG;
  for (int i = 0; i < CNERF_MAX_TENSORS; ++i)
    if (grads0->p[i] && grads0->p[i] == grads1->p[i]) return CNERF_E_ARG;
  return wgrad_two(net0, B0, S0, stash0, workspace0, grads0, net1, B1, S1, stash1, workspace1, grads1, accumulate, stream, 0,
                   live_rays, first_ray0, first_ray1);
}

// cnerf_mlp_bwd_pair of two levels of ONE ray batch padded to a fixed capacity (B0 == B1 rays) whose LIVE row count sits in device
// memory: both levels' dgrad tiles and wgrad point ranges stop at live_rays * S of their level
extern "C" int cnerf_mlp_bwd_pair_live(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                                       const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                                       const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                                       const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate,
                                       const int32_t* live_rays, int64_t first_ray0, int64_t first_ray1, void* stream) {
  if (!grads0 || !grads1 || !live_rays || B0 != B1) return CNERF_E_ARG;
  for (int i = 0; i < CNERF_MAX_TENSORS; ++i)
    if (grads0->p[i] && grads0->p[i] == grads1->p[i]) return CNERF_E_ARG;
  int rc = dgrad_pair(net0, packed0, d_raw0, B0, S0, stash0, workspace0, net1, packed1, d_raw1, B1, S1, stash1, workspace1, live_rays,
                      stream, first_ray0, first_ray1);
  if (rc) return rc;
  return wgrad_two(net0, B0, S0, stash0, workspace0, grads0, net1, B1, S1, stash1, workspace1, grads1, accumulate, stream, 0,
                   live_rays, first_ray0, first_ray1);
}
