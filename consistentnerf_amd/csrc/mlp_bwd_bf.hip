// OPT-IN "bf16x3" dgrad of the fused encoding + MLP (second bench line only; the default training path is mlp_bwd.hip, exact
// fp32): the activation-gradient chain of NeRF.forward (autograd of H:107-130) with its GEMMs on v_mfma_f32_32x32x16_bf16,
// every operand split into THREE bf16 planes (6 cross terms, fp32 accumulation: error per product ~2^-23, fp32-equivalent).
//
// Same walk as mlp_dgrad_k — one wave64 owns 32 points, dZ_{l-1}^T = relu'(h_{l-1}) (.) (W_l^T . dZ_l^T), the previous gradient's
// accumulator registers are the B operand of the next GEMM, ReLU masks from the sign-bit words of the stash, every dZ_l written
// to the tile-major gradient workspace G while it is the B operand — with the machinery of mlp_fwd_bf.hip: the four waves of a
// workgroup (128 points) share ONE stream of pre-split transposed weight panels through the LDS ring, the B planes are split on
// the VALU in the gaps between MFMAs.  The sigma / rgb heads stay fp32 on the VALU.  Output: the same G the fp32 wgrad reads.
#include "mlp_bf_common.hpp"

int cn_make_geom(const cnerf_net* net, NetGeom* g);

namespace {

struct BfBwdLevel {
  const unsigned char* pk;   // cnerf_pack_weights_bf(net, params, 3, ...)
  const float* d_raw;
  const float* stash;
  float* G;
  int64_t M, Mp;
};

struct BfBwdArgs {
  NetGeom g;
  BfGeom b;
  BfBwdLevel lv[2];
  unsigned nb0;              // workgroups [0, nb0) walk level 0, the rest level 1 (coarse and fine network of one step)
};

template <int NT>
__global__ __launch_bounds__(256) void mlp_dgrad_bfs_k(BfBwdArgs args_by_value) {
  constexpr int NP = 3, W = NT * 32, NTH = NT / 2;
  constexpr int MD = (NT + 1) / 2, MDV = (NTH + 1) / 2;
  (void)args_by_value;
  const CN_CONST BfBwdArgs& args = *(const CN_CONST BfBwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const CN_CONST NetGeom& g = args.g;
  const CN_CONST BfGeom& bg = args.b;
  const unsigned nb0 = args.nb0;
  const bool second = blockIdx.x >= nb0;
  const CN_CONST BfBwdLevel& a = args.lv[second ? 1 : 0];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [4 ring slots]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // a wave past the last point still moves its share of the panels and meets the barriers: clamped points, empty resources
  const int64_t p0 = ((int64_t)(blockIdx.x - (second ? nb0 : 0u)) * 4 + w) * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (a.M - p0 > 0 ? (int)(a.M - p0) : 0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
  const bool valid = p < a.M;
  const BfPanel P{make_rsrc(a.pk, (unsigned)bg.total), lane * 16};
  const rsrc_t srs = make_rsrc(nvalid > 0 ? a.stash + p0 * g.s_rows : nullptr, nvalid > 0 ? (unsigned)(32 * g.s_rows * 4) : 0u);
  const rsrc_t grs = make_rsrc(nvalid > 0 ? a.G + p0 * g.g_rows : nullptr, nvalid > 0 ? (unsigned)(32 * g.g_rows * 4) : 0u);
  const int gvo = valid ? m * 32 + hh * 16 : TM_OOB;
  const int smo = valid ? m * 32 + hh * MD * 4 : TM_OOB;
  Ring<NT, NP> R;
  {
    const unsigned long long ba = (unsigned long long)a.pk;
    R.rs = i32x4{__builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((ba >> 32) & 0xffff)),
                 __builtin_amdgcn_readfirstlane((int)bg.total), 0x00027000};
  }
  R.lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds_raw);
  R.ring = lds_raw;
  R.w = w;
  R.lane16 = lane * 16;
  R.rd16 = lane * 16;
  // start the panel stream before anything else: K-steps 0 and 1 of views_linears^T
  R.dma((int)bg.pt_views, 0, 0);
  R.dma((int)bg.pt_views, 1, 1);

  const float4 d = *reinterpret_cast<const float4*>(a.d_raw + pc * 4);
  const float dc[4] = {d.x, d.y, d.z, d.w};
  unsigned bv[MDV], bits[MD];
  load_bits<MDV>(srs, valid ? m * 32 + hh * MDV * 4 : TM_OOB, tm_col(g.s_mask + g.s_mb[g.D]), bv);
  if (hh == 0) buf_store(grs, valid ? m * 32 : TM_OOB, tm_col(g.g_out), f32x4{d.x, d.y, d.z, d.w});
  // rgb_linear^T on the VALU (fp32), masked by the view-branch ReLU -> dZv (C-layout registers)
  f32x16 V[NTH];
  {
    f32x4 wq[3][NTH][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int t = 0; t < NTH; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wq[c][t][q] = buf_load(P.rs, hh * 16, (int)bg.v_rgb + (c * (W / 2) + 32 * t + 8 * q) * 4);
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
  pin<NTH>(V);
  R.template publish<Ring<NT, NP>::PW>();      // K-step 0 of views_linears^T has landed everywhere

  f32x16 X[NT], Y[NT];
  // dF = views_linears^T (feature columns; gamma(d) needs no gradient) . dZv — no mask (feature_linear is linear, H:118)
  gemm_ring_reg<NTH, NT, NT, NP, false, StashStores<NTH>, 2, true>(X, V, R, (int)bg.pt_views, (int)bg.pt_feat,
                                                                   StashStores<NTH>{V, grs, gvo, tm_col(g.g_hv)});
  pin<NT>(X);
  // dZ_{D-1} = relu'(h_{D-1}) * (feature_linear^T . dF + alpha_linear^T . dsigma): the sigma term seeds the accumulators
  load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[g.D - 1]), bits);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wq = buf_load(P.rs, hh * 16, (int)bg.v_alpha + (32 * t + 8 * q) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) Y[t][4 * q + j] = wq[j] * dc[3];
    }
  pin<NT>(Y);      // the head-weight loads retire (compiler waitcnt) before the ring's own vmcnt bookkeeping resumes
  gemm_ring_reg<NT, NT, NT, NP, false, StashStores<NT>, 2, false>(Y, X, R, (int)bg.pt_feat,
                                                                  g.D > 1 ? (int)bg.pt_trunk[g.D - 1] : -1,
                                                                  StashStores<NT>{X, grs, gvo, tm_col(g.g_feat)});
  mask_bits<NT>(Y, bits);
  pin<NT>(Y);
  // trunk: dZ_{l-1} = relu'(h_{l-1}) * (W_l^T . dZ_l) (the gamma(x) columns of the skip layer get no gradient); dZ_l goes out
  // to the workspace while it is the B operand of this GEMM.  X / Y alternate as input and output.
  auto layer = [&](f32x16 (&In)[NT], f32x16 (&Out)[NT], int l) __attribute__((always_inline)) {
    load_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[l - 1]), bits);
    gemm_ring_reg<NT, NT, NT, NP, false, StashStores<NT>, 2, true>(Out, In, R, (int)bg.pt_trunk[l], l > 1 ? (int)bg.pt_trunk[l - 1] : -1,
                                                                   StashStores<NT>{In, grs, gvo, tm_col(g.g_z[l])});
    mask_bits<NT>(Out, bits);
    pin<NT>(Out);
  };
  int l = g.D - 1;
  for (; l >= 2; l -= 2) {
    layer(Y, X, l);
    layer(X, Y, l - 1);
  }
  if (l == 1) {
    layer(Y, X, 1);
    store_tiles<NT>(X, grs, gvo, tm_col(g.g_z[0]));
  } else {
    store_tiles<NT>(Y, grs, gvo, tm_col(g.g_z[0]));
  }
}

template <int NT>
int launch(const BfBwdArgs& a, int nlev, hipStream_t st) {
  const size_t lds = (size_t)4 * Ring<NT, 3>::SLOT;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CNERF_E_NODEVICE;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_bfs_k<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return (int)hipGetLastError();
    attr_set[dev] = true;
  }
  unsigned grid = 0;
  for (int i = 0; i < nlev; ++i) {
    const BfBwdLevel& L = a.lv[i];
    grid += (unsigned)cn_div_up(L.M, 128);
    if (L.Mp > L.M) {   // last gradient tile row holds padding points: the kernel drops their stores, wgrad reads them
      hipError_t e = hipMemsetAsync(L.G + (L.Mp - 32) * a.g.g_rows, 0, (size_t)32 * a.g.g_rows * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  hipLaunchKernelGGL((mlp_dgrad_bfs_k<NT>), dim3(grid), dim3(256), lds, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

int dispatch(const BfBwdArgs& a, int nlev, hipStream_t st) {
  if (a.g.in_chp != 64 || a.g.dir_chp != 32) return CNERF_E_UNSUPPORTED;
  switch (a.g.NT) {
    case 4: return launch<4>(a, nlev, st);
    case 8: return launch<8>(a, nlev, st);
  }
  return CNERF_E_UNSUPPORTED;
}

}  // namespace

extern "C" int cnerf_mlp_dgrad_bf(const cnerf_net* net, const void* packed_bf, const float* d_raw, int64_t B, int S,
                                  const float* stash, float* workspace, void* stream) {
  BfBwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if ((rc = make_bf_geom(a.g, 3, &a.b))) return rc;
  if (!packed_bf || !d_raw || !stash || !workspace || B < 0 || S <= 0) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.lv[0] = BfBwdLevel{static_cast<const unsigned char*>(packed_bf), d_raw, stash, workspace, B * S, cn_round_up(B * S, 32)};
  a.lv[1] = a.lv[0];
  a.nb0 = (unsigned)cn_div_up(B * S, 128);
  return dispatch(a, 1, cn_stream(stream));
}

// Two independent networks of the same architecture (coarse and fine of one training step) in ONE grid; different
// architectures take two launches.
extern "C" int cnerf_mlp_dgrad_bf_pair(const cnerf_net* net0, const void* packed_bf0, const float* d_raw0, int64_t B0, int S0,
                                       const float* stash0, float* workspace0, const cnerf_net* net1, const void* packed_bf1,
                                       const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1,
                                       void* stream) {
  BfBwdArgs a;
  int rc = cn_make_geom(net0, &a.g);
  if (rc) return rc;
  if ((rc = make_bf_geom(a.g, 3, &a.b))) return rc;
  if (!packed_bf0 || !d_raw0 || !stash0 || !workspace0 || !packed_bf1 || !d_raw1 || !stash1 || !workspace1 || B0 < 0 || B1 < 0 ||
      S0 <= 0 || S1 <= 0)
    return CNERF_E_ARG;
  const int64_t M0 = B0 * S0, M1 = B1 * S1;
  const bool same = net0->D == net1->D && net0->W == net1->W && net0->multires == net1->multires &&
                    net0->multires_views == net1->multires_views && net0->use_viewdirs == net1->use_viewdirs &&
                    net0->output_ch == net1->output_ch && net0->skip == net1->skip;
  if (same && M0 > 0 && M1 > 0) {
    a.lv[0] = BfBwdLevel{static_cast<const unsigned char*>(packed_bf0), d_raw0, stash0, workspace0, M0, cn_round_up(M0, 32)};
    a.lv[1] = BfBwdLevel{static_cast<const unsigned char*>(packed_bf1), d_raw1, stash1, workspace1, M1, cn_round_up(M1, 32)};
    a.nb0 = (unsigned)cn_div_up(M0, 128);
    return dispatch(a, 2, cn_stream(stream));
  }
  if ((rc = cnerf_mlp_dgrad_bf(net0, packed_bf0, d_raw0, B0, S0, stash0, workspace0, stream))) return rc;
  return cnerf_mlp_dgrad_bf(net1, packed_bf1, d_raw1, B1, S1, stash1, workspace1, stream);
}
