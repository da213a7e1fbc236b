// Backward of the fused encoding+MLP (autograd of run_network R:37-52 / NeRF.forward H:107-130).
// Three launches:
//   1. dgrad (this file): one wave64 per 32 points walks the network backwards with the TRANSPOSED weight
//      panels as the MFMA A operand and the gradient tile in LDS as B; it writes the gradient w.r.t. every
//      layer's pre-activation, dZ_l, into the point-major gradient workspace G[Mp][g_rows].
//   2. wgrad (wgrad.hip): NT GEMMs contracted over points, dW_l = dZ_l^T . H_{l-1}, split over point ranges.
//   3. a fixed-order reduction of the split partials into the parameter gradients (deterministic).
// ReLU masks are re-derived from the forward stash (H > 0  <=>  pre-activation > 0).
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

// Prefetch the stash block whose ReLU decides the mask (C-layout features of this lane = 4*NTO dwordx4 loads off
// the point's stash row); issued BEFORE the GEMM that produces the gradient so the HBM latency hides under it.
template <int NTO>
__device__ __forceinline__ void load_rows(f32x16 (&h)[NTO], const float* __restrict__ sp, int col) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sp + col + 32 * t + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[t][4 * q + j] = v[j];
    }
}

// The same loads as a `side` functor of gemm_pipe (one 16-byte load behind an MFMA each): slot i -> tile i/4, quad i%4.
template <int NTO>
struct RowLoader {
  f32x16 (&h)[NTO];
  const float* __restrict__ src;   // sp + col
  __device__ __forceinline__ void operator()(int i) const {
    const int t = i >> 2, q = i & 3;
    if (t < NTO) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 32 * t + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[t][4 * q + j] = v[j];
    }
  }
};

// mask (H>0) and park: acc <- acc * [h > 0]; the masked gradient goes to the LDS tile (B operand of the next
// transposed GEMM) and, straight from the registers, to this point's row of the point-major gradient workspace
// (`gp` = row + 4*hh, block column `col`): one 16-byte store per 4 consecutive features.  (Measured: routing the
// stores through the LDS tile for 1 KiB-coalesced writes is SLOWER here — the ds_read -> store chain is exposed
// latency on a one-wave-per-SIMD kernel, while L2 write-combines the 32-byte lane-pair pieces anyway.)
// Padding points store zeros.
template <int W, int NTO, bool MASK>
__device__ __forceinline__ void mask_park(f32x16 (&acc)[NTO], const f32x16 (&h)[NTO], float* Hs,
                                          float* __restrict__ gp, int col, bool valid, int m, int hh) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = acc[t][4 * q + j];
        if (MASK) x = h[t][4 * q + j] > 0.f ? x : 0.f;
        v[j] = x;
      }
      *reinterpret_cast<f32x4*>(Hs + hs_off<W>(m, 8 * t + 2 * q + hh)) = v;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(gp + col + 32 * t + 8 * q) = valid ? v : z;
    }
}

template <int NT, bool VD>
__global__ __launch_bounds__(64) void mlp_dgrad_k(BwdArgs a) {
  constexpr int W = NT * 32;
  constexpr int NTH = NT / 2 > 0 ? NT / 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float Hs[];
  const NetGeom& g = a.g;
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p = (int64_t)blockIdx.x * 32 + m;
  const bool valid = p < a.M;
  const int64_t pc = valid ? p : a.M - 1;
  const float* pk = a.packed;
  const APanel AP{make_rsrc(a.packed, (unsigned)(g.total * 4)), (m * 8 + 4 * hh) * 4};
  const float* const sp = a.stash + p * g.s_rows + 4 * hh;   // this point's stash row (+ this half's features)
  float* const gp = a.G + p * g.g_rows + 4 * hh;             // this point's gradient row (+ this half's features)
  f32x16 acc[NT];
  f32x16 hm[NT];     // prefetched stash features for the next ReLU mask
  f32x4 a0[NT], a1[NT];   // A-operand sets (even / odd K-groups) of the current / next transposed panel
  CN_TINIT(1)

  if (VD) {
    const float4 d = *reinterpret_cast<const float4*>(a.d_raw + pc * 4);
    const float dc[4] = {d.x, d.y, d.z, d.w};
    // rgb_linear^T on the VALU, masked by the view-branch ReLU -> dZv (C-layout registers)
    f32x16 accv[NTH];
    f32x16 hv[NTH];
    load_rows<NTH>(hv, sp, g.s_hv);
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = 32 * t + 8 * q + 4 * hh;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_rgb + (int64_t)c * g.Wh + n0);
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] += w[j] * dc[c];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) accv[t][4 * q + j] = s[j];
      }
    CN_T(0)
    mask_park<W, NTH, true>(accv, hv, Hs, gp, g.g_hv, valid, m, hh);
    __builtin_amdgcn_wave_barrier();
    CN_T(3)
    a_prefetch<NT>(a0, a1, AP, (int)g.t_views, W, g.Wh / 8 - 1);
    if (hh == 0) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(a.G + p * g.g_rows + g.g_out) = valid ? d : z4;
    }
    // views_linears^T (feature columns only; gamma(d) needs no gradient) -> dF
    zero_acc<NT>(acc);
    CN_T(4)
    gemm_pipe<W, NT, false>(acc, a0, a1, AP, (int)g.t_views, W, g.Wh / 8, Hs, m, hh);
    a_prefetch<NT>(a0, a1, AP, (int)g.t_feat, W, W / 8 - 1);
    __builtin_amdgcn_wave_barrier();
    CN_T(2)
    mask_park<W, NT, false>(acc, hm, Hs, gp, g.g_feat, valid, m, hh);
    __builtin_amdgcn_wave_barrier();
    CN_T(3)
    // feature_linear^T . dF  +  alpha_linear^T . dsigma, masked by the last trunk ReLU -> dZ_{D-1}
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_alpha + 32 * t + 8 * q + 4 * hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = w[j] * dc[3];
      }
    CN_T(4)
    gemm_pipe<W, NT, false>(acc, a0, a1, AP, (int)g.t_feat, W, W / 8, Hs, m, hh,
                            RowLoader<NT>{hm, sp + g.s_h[g.D - 1]});
    CN_T(2)
  } else {
    load_rows<NT>(hm, sp, g.s_h[g.D - 1]);
    float dc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dc[c] = c < g.out_ch ? a.d_raw[pc * g.out_ch + c] : 0.f;
    // output_linear^T on the VALU
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = 32 * t + 8 * q + 4 * hh;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < g.out_ch) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_out + (int64_t)c * W + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += w[j] * dc[c];
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = s[j];
      }
    if (hh == 0)
      for (int c = 0; c < g.out_ch; ++c) a.G[p * g.g_rows + g.g_out + c] = valid ? dc[c] : 0.f;
  }
  if (g.D > 1) a_prefetch<NT>(a0, a1, AP, (int)g.t_trunk[g.D - 1], W, W / 8 - 1);
  __builtin_amdgcn_wave_barrier();
  CN_T(0)
  mask_park<W, NT, true>(acc, hm, Hs, gp, g.g_z[g.D - 1], valid, m, hh);
  __builtin_amdgcn_wave_barrier();
  CN_T(3)
  // trunk: dZ_{l-1} = relu'(.) * W_l^T dZ_l   (the gamma(x) columns of the skip layer get no gradient)
  for (int l = g.D - 1; l >= 1; --l) {
    zero_acc<NT>(acc);
    CN_T(4)
    gemm_pipe<W, NT, false>(acc, a0, a1, AP, (int)g.t_trunk[l], W, W / 8, Hs, m, hh,
                            RowLoader<NT>{hm, sp + g.s_h[l - 1]});
    if (l > 1) a_prefetch<NT>(a0, a1, AP, (int)g.t_trunk[l - 1], W, W / 8 - 1);
    __builtin_amdgcn_wave_barrier();
    CN_T(2)
    mask_park<W, NT, true>(acc, hm, Hs, gp, g.g_z[l - 1], valid, m, hh);
    __builtin_amdgcn_wave_barrier();
    CN_T(3)
  }
  CN_TEND
}

template <int NT>
int launch(const BwdArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  const size_t lds = (size_t)NT * 32 * 32 * sizeof(float);
  if (a.g.viewdirs) hipLaunchKernelGGL((mlp_dgrad_k<NT, true>), dim3(grid), dim3(64), lds, st, a);
  else hipLaunchKernelGGL((mlp_dgrad_k<NT, false>), dim3(grid), dim3(64), lds, st, a);
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
