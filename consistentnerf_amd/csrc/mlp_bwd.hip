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
                    int nsplit, const cnerf_ptrs* grads, int accumulate, hipStream_t st);
int cn_wgrad_nsplit(int64_t Mp);
int64_t cn_param_floats(const NetGeom& g);

namespace {

struct BwdArgs {
  NetGeom g;
  const float* packed;
  const float* d_raw;
  const float* stash;
  float* G;
  int64_t M, Mp;
};

template <int NT, bool VD>
__global__ __launch_bounds__(64) void mlp_dgrad_k(BwdArgs a) {
  constexpr int W = NT * 32;
  constexpr int NTH = NT / 2 > 0 ? NT / 2 : 1;
  constexpr int MD = (NT + 1) / 2, MDV = (NTH + 1) / 2;
  const NetGeom& g = a.g;
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (int)(a.M - p0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
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
    a_prefetch3<NT>(A, AP, (int)g.t_feat, W, W / 8 - 1);
    load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[g.D - 1]), bits);
    {
      f32x4 wq[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[t][q] = buf_load(AP.rs, hh * 16, (int)(g.v_alpha + 32 * t + 8 * q) * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) Y[t][4 * q + j] = wq[t][q][j] * dc[3];
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
int launch(const BwdArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  if (a.Mp > a.M) {   // last gradient tile row holds padding points: the kernel drops their stores, wgrad reads them
    hipError_t e = hipMemsetAsync(a.G + (a.Mp - 32) * a.g.g_rows, 0, (size_t)32 * a.g.g_rows * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  if (a.g.viewdirs) hipLaunchKernelGGL((mlp_dgrad_k<NT, true>), dim3(grid), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((mlp_dgrad_k<NT, false>), dim3(grid), dim3(64), 0, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
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

extern "C" int cnerf_mlp_dgrad(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                               const float* stash, float* workspace, void* stream) {
  BwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !d_raw || !stash || !workspace || B < 0 || S <= 0) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.packed = packed; a.d_raw = d_raw; a.stash = stash; a.G = workspace;
  a.M = B * S; a.Mp = cn_round_up(a.M, 32);
  hipStream_t st = cn_stream(stream);
  switch (a.g.NT) {
    case 2: return launch<2>(a, st);
    case 4: return launch<4>(a, st);
    case 8: return launch<8>(a, st);
  }
  return CNERF_E_UNSUPPORTED;
}

extern "C" int cnerf_mlp_wgrad(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace,
                               const cnerf_ptrs* grads, int accumulate, void* stream) {
  NetGeom g;
  int rc = cn_make_geom(net, &g);
  if (rc) return rc;
  if (!stash || !workspace || !grads || B < 0 || S <= 0) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  const int64_t M = B * S, Mp = cn_round_up(M, 32);
  const int nsplit = cn_wgrad_nsplit(Mp);
  float* partials = workspace + (int64_t)g.g_rows * Mp;
  return cn_wgrad_launch(g, stash, workspace, M, Mp, partials, nsplit, grads, accumulate, cn_stream(stream));
}

extern "C" int cnerf_mlp_bwd(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                             const float* stash, float* workspace, const cnerf_ptrs* grads, int accumulate,
                             void* stream) {
  if (!grads) return CNERF_E_ARG;
  int rc = cnerf_mlp_dgrad(net, packed, d_raw, B, S, stash, workspace, stream);
  if (rc) return rc;
  return cnerf_mlp_wgrad(net, B, S, stash, workspace, grads, accumulate, stream);
}
