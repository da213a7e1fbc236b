// Fused positional-encoding + NeRF MLP forward on gfx950 matrix cores.
// Replaces run_network (R:37-52), Embedder.embed (H:15-63) and NeRF.forward (H:107-130): the 90-wide
// encodings and every [M,256] activation of the reference never reach HBM (inference), or reach it
// exactly once as the training stash.
//
// Mapping.  A workgroup of two wave64 owns 32 points and walks them through the whole network; 4 workgroups per
// CU = 2 waves per SIMD.  Every layer is computed TRANSPOSED, Out^T[N x 32] = W[N x K] . H^T[K x 32], with
// v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain); wave w produces output tiles
// [w*NT/2, (w+1)*NT/2):
//   A operand  = weights, lane (i = lane&31, hh = lane>>5) holds W[n0+i][k + hh'] — one 16-byte load of
//                the packed panel (common.hpp) feeds 4 consecutive MFMAs; a wave reads 1 KiB contiguous.
//   B operand  = activations of the 32 points, lane (m = lane&31, hh) holds H^T[k][m]; read from the workgroup's
//                LDS tile Hs[m][k] (16-byte chunks XOR-swizzled by m&15: conflict-free b128).
//   D (C-layout) lane (m, hh) holds rows n = 32t + 8(r>>2) + 4hh + (r&3): 4 consecutive n per float4,
//                so ReLU'd accumulators go back to Hs with ds_write_b128 and straight into the next layer.
// Two workgroup barriers per layer (tile fully read -> overwrite -> fully written); weights (2.4 MB/net) stay
// L2-resident and are streamed once per 32 points; MFMA-bound by construction (593 920 MAC per point at
// D=8/W=256 incl. K padding).
#include "mlp_common.hpp"

#include "timing.hpp"

namespace {

struct FwdArgs {
  NetGeom g;
  const float* packed;
  const float* pts;
  const float* rays;
  const float* dirs;
  const float* z;
  const float* emb;   // pre-embedded inputs [M, in_ch + dir_ch] (NeRF.forward surface) or nullptr
  float* raw;
  float* stash;
  int64_t M, Mp;
  int S, rs;
};

// gamma(x) channels of one point into Hs[m][0..chp), split over the workgroup's two waves (w) and the two
// half-waves (hh): wave w takes the frequencies l = w, w+2, ...; hh=0 lanes their sines, hh=1 their cosines;
// wave 0 / hh 0 the identity channels, wave 1 / hh 1 the zero padding.
// Channel order H:24-45: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(..)].  `srow` = this point's stash
// row + block column (training) or nullptr.
template <int W>
__device__ __forceinline__ void encode(float* Hs, const float (&x)[3], int L, int ch, int chp, int w, int m, int hh,
                                       float* __restrict__ srow, bool valid, const float* __restrict__ pre = nullptr) {
  auto put = [&](int k, float v) {
    Hs[hs_off<W>(m, k >> 2) + (k & 3)] = v;
    if (srow != nullptr) srow[k] = valid ? v : 0.f;
  };
  if (pre != nullptr) {   // NeRF.forward(x) on an already-embedded batch (H:107-109): copy this point's channels
    for (int k = 2 * w + hh; k < chp; k += 4) put(k, k < ch ? pre[k] : 0.f);
    return;
  }
  if (w == 0 && hh == 0) {
    put(0, x[0]); put(1, x[1]); put(2, x[2]);
  }
  if (w == 1 && hh == 1) {
    for (int k = ch; k < chp; ++k) put(k, 0.f);
  }
  float f = w ? 2.f : 1.f;
  for (int l = w; l < L; l += 2) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float arg = x[d] * f;
      put(3 + 6 * l + 3 * hh + d, hh ? cosf(arg) : sinf(arg));
    }
    f *= 4.f;
  }
}

template <int NT, bool VD>
__global__ __launch_bounds__(128, 2) void mlp_fwd_k(FwdArgs a) {
  constexpr int W = NT * 32;
  constexpr int NTW = NT / 2;                       // trunk / feature tiles per wave
  constexpr int NTH = NT / 2;                       // view-branch tiles (W/2 wide)
  constexpr int NTHW = NTH / 2 > 0 ? NTH / 2 : 1;   // ... per wave (W=64: one tile, wave 0 only)
  extern __shared__ __attribute__((aligned(16))) float Hs[];   // [32][W] tile + 128 floats of head scratch
  const NetGeom& g = a.g;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hh = lane >> 5;
  const int64_t p = (int64_t)blockIdx.x * 32 + m;
  const bool valid = p < a.M;
  const int64_t pc = valid ? p : a.M - 1;
  const int64_t ray = pc / a.S;
  const float* pk = a.packed;
  const int t0 = w * NTW;

  CN_TINIT(2)
  float x[3] = {0.f, 0.f, 0.f};
  const float* const pre = a.emb != nullptr ? a.emb + pc * (g.in_ch + g.dir_ch) : nullptr;
  if (pre != nullptr) {
  } else if (a.pts != nullptr) {
    x[0] = a.pts[pc * 3 + 0]; x[1] = a.pts[pc * 3 + 1]; x[2] = a.pts[pc * 3 + 2];
  } else {
    const float* r = a.rays + ray * a.rs;
    const float zz = a.z[pc];
    x[0] = r[0] + r[3] * zz; x[1] = r[1] + r[4] * zz; x[2] = r[2] + r[5] * zz;   // R:384 (no FMA contraction)
  }
  // training: this point's stash row (point-major [Mp][s_rows]); padding points p in [M, Mp) are stored as zeros
  float* const srow = a.stash != nullptr ? a.stash + p * g.s_rows : nullptr;
  float* const sp = srow != nullptr ? srow + 4 * hh : nullptr;

  encode<W>(Hs, x, g.L, g.in_ch, g.in_chp, w, m, hh, srow != nullptr ? srow + g.s_enc : nullptr, valid, pre);
  CN_T(0)
  __syncthreads();
  CN_T(1)

  // Per layer: GEMM -> [queue the next panel's first A groups] -> barrier (tile fully read) -> park (ReLU, LDS,
  // stash) -> barrier -> next GEMM.  Biases ride on the panels (gemm_run<BIAS>), accumulators start at zero.
  f32x16 acc[NTW];
  f32x16 accs[NTW];   // gamma(x) part (+ bias) of the skip layer, computed while gamma(x) is still in LDS
  Ring<NTW> R;
  zero_acc<NTW>(acc);
  gemm_seg<W, NTW, true>(acc, pk + g.f_l0 + t0 * 256, W, g.in_chp / 8, Hs, m, hh);
#if !(defined(CN_EXP) && (CN_EXP & 8))   // ablation: no skip partial (wrong results, frees 64 registers)
  if (g.skip >= 0) {
    zero_acc<NTW>(accs);
    gemm_seg<W, NTW, true>(accs, pk + g.f_skip + t0 * 256, W, g.in_chp / 8, Hs, m, hh);
  }
#endif
  CN_T(2)
  for (int l = 1; l < g.D; ++l) {
    const bool sk = l == g.skip + 1;
    const float* panel = pk + g.f_trunk[l] + t0 * 256;
    ring_start<NTW>(R, panel, W, sk ? W / 8 - 1 : W / 8, m, hh);
    __syncthreads();                                 // both waves finished reading the tile
    CN_T(1)
    park<W, NTW, true>(acc, Hs, true, t0, m, hh, sp, g.s_h[l - 1], valid);
    CN_T(3)
    __syncthreads();
    CN_T(1)
    if (sk) {
#if defined(CN_EXP) && (CN_EXP & 8)
      zero_acc<NTW>(acc);
#else
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = accs[t];
#endif
      gemm_run<W, NTW, false>(acc, R, panel, W, W / 8, Hs, m, hh);
    } else {
      zero_acc<NTW>(acc);
      gemm_run<W, NTW, true>(acc, R, panel, W, W / 8, Hs, m, hh);
    }
    CN_T(2)
  }
  if (VD) ring_start<NTW>(R, pk + g.f_feat + t0 * 256, W, W / 8, m, hh);
  __syncthreads();
  CN_T(1)
  park<W, NTW, true>(acc, Hs, true, t0, m, hh, sp, g.s_h[g.D - 1], valid);
  CN_T(3)
  __syncthreads();
  CN_T(1)

  if (!VD) {
    // output_linear (H:127-128) on the VALU of wave 0: out[c] = b[c] + sum_k Wo[c][k] h[k]; each half-wave sums
    // half the chunks
    if (w == 0) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = 0.f;
      for (int i = 0; i < W / 8; ++i) {
        const int ck = 2 * i + hh;
        const f32x4 h = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, ck));
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < g.out_ch) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(pk + g.v_out + (int64_t)c * W + 4 * ck);
            o[c] += h[0] * wv[0] + h[1] * wv[1] + h[2] * wv[2] + h[3] * wv[3];
          }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] += __shfl_xor(o[c], 32, 64);
      if (valid && hh == 0)
        for (int c = 0; c < g.out_ch; ++c) a.raw[p * g.out_ch + c] = o[c] + pk[g.b_out + c];
    }
    CN_T(4)
    CN_TEND
    return;
  } else {
    // sigma head (alpha_linear, H:117) on the VALU of wave 0 while the trunk output is in LDS
    float sig = 0.f;
    if (w == 0) {
      for (int i = 0; i < W / 8; ++i) {
        const int ck = 2 * i + hh;
        const f32x4 h = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, ck));
        const f32x4 wv = *reinterpret_cast<const f32x4*>(pk + g.v_alpha + 4 * ck);
        sig += h[0] * wv[0] + h[1] * wv[1] + h[2] * wv[2] + h[3] * wv[3];
      }
      sig += __shfl_xor(sig, 32, 64);
      sig += pk[g.b_alpha];
    }
    CN_T(4)
    // feature_linear (H:118), no activation
    zero_acc<NTW>(acc);
    gemm_run<W, NTW, true>(acc, R, pk + g.f_feat + t0 * 256, W, W / 8, Hs, m, hh);
    CN_T(2)
    __syncthreads();                                 // trunk output dead
    CN_T(1)
    // gamma(viewdir) overwrites the trunk tile; its share of views_linears first
    float v[3] = {0.f, 0.f, 0.f};
    if (pre == nullptr) {
      const float* dsrc = a.dirs != nullptr ? a.dirs + ray * 3 : a.rays + ray * a.rs + (a.rs - 3);
      v[0] = dsrc[0]; v[1] = dsrc[1]; v[2] = dsrc[2];
    }
    encode<W>(Hs, v, g.Ld, g.dir_ch, g.dir_chp, w, m, hh, srow != nullptr ? srow + g.s_denc : nullptr, valid,
              pre != nullptr ? pre + g.in_ch : nullptr);
    CN_T(0)
    __syncthreads();
    CN_T(1)
    const int t0v = w * NTHW;
    const bool vact = t0v < NTH;                     // wave-uniform
    f32x16 accv[NTHW];
    if (vact) {
      zero_acc<NTHW>(accv);
      gemm_seg<W, NTHW, true>(accv, pk + g.f_viewsd + t0v * 256, g.Wh, g.dir_chp / 8, Hs, m, hh);
    }
    CN_T(2)
    __syncthreads();                                 // gamma(d) dead
    CN_T(1)
    park<W, NTW, false>(acc, Hs, true, t0, m, hh, sp, g.s_feat, valid);
    CN_T(3)
    __syncthreads();
    CN_T(1)
    float o[3] = {0.f, 0.f, 0.f};
    if (vact) {
      gemm_seg<W, NTHW, false>(accv, pk + g.f_views + t0v * 256, g.Wh, W / 8, Hs, m, hh);
      CN_T(2)
      park<W, NTHW, true>(accv, Hs, false, t0v, m, hh, sp, g.s_hv, valid);   // ReLU in registers (+ stash)
      // rgb_linear (H:125) straight from the accumulators: lane holds n = 32t + 8q + 4hh + j
#pragma unroll
      for (int t = 0; t < NTHW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(pk + g.v_rgb + (int64_t)c * g.Wh + 32 * (t0v + t) +
                                                             8 * q + 4 * hh);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[c] += accv[t][4 * q + j] * wv[j];
          }
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] += __shfl_xor(o[c], 32, 64);
    }
    float* scratch = Hs + 32 * W;                    // 32 points x 4 floats, outside the tile
    if (w == 1 && hh == 0) {
      scratch[4 * m + 0] = o[0]; scratch[4 * m + 1] = o[1]; scratch[4 * m + 2] = o[2];
    }
    CN_T(4)
    __syncthreads();
    CN_T(1)
    if (w == 0 && hh == 0 && valid) {
      *reinterpret_cast<float4*>(a.raw + p * 4) =
          make_float4(o[0] + scratch[4 * m + 0] + pk[g.b_rgb + 0], o[1] + scratch[4 * m + 1] + pk[g.b_rgb + 1],
                      o[2] + scratch[4 * m + 2] + pk[g.b_rgb + 2], sig);
    }
    CN_T(4)
    CN_TEND
  }
}

template <int NT>
int launch(const FwdArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  const size_t lds = (size_t)(NT * 32 * 32 + 128) * sizeof(float);
  if (a.g.viewdirs) hipLaunchKernelGGL((mlp_fwd_k<NT, true>), dim3(grid), dim3(128), lds, st, a);
  else hipLaunchKernelGGL((mlp_fwd_k<NT, false>), dim3(grid), dim3(128), lds, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

}  // namespace

#ifdef CN_TIMING
CN_TIMING_ACCESSOR(cnerf_debug_timing)
#endif

extern "C" int cnerf_mlp_fwd(const cnerf_net* net, const float* packed, const float* pts, const float* rays,
                             int ray_stride, const float* dirs, const float* z, int64_t B, int S, float* raw,
                             float* stash, void* stream) {
  FwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !raw || B < 0 || S <= 0) return CNERF_E_ARG;
  if (!pts && (!rays || !z || ray_stride < 8)) return CNERF_E_ARG;
  if (a.g.viewdirs && !dirs && (!rays || ray_stride < 11)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.packed = packed; a.pts = pts; a.rays = rays; a.dirs = dirs; a.z = z; a.emb = nullptr; a.raw = raw;
  a.stash = stash;
  a.M = B * S; a.Mp = cn_round_up(a.M, 32); a.S = S; a.rs = ray_stride;
  switch (a.g.NT) {
    case 2: return launch<2>(a, cn_stream(stream));
    case 4: return launch<4>(a, cn_stream(stream));
    case 8: return launch<8>(a, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
}

extern "C" int cnerf_mlp_fwd_embedded(const cnerf_net* net, const float* packed, const float* x_embedded, int64_t M,
                                      float* raw, float* stash, void* stream) {
  FwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !x_embedded || !raw || M < 0) return CNERF_E_ARG;
  if (M == 0) return CNERF_OK;
  a.packed = packed; a.pts = nullptr; a.rays = nullptr; a.dirs = nullptr; a.z = nullptr; a.emb = x_embedded;
  a.raw = raw; a.stash = stash;
  a.M = M; a.Mp = cn_round_up(M, 32); a.S = 1; a.rs = 0;
  switch (a.g.NT) {
    case 2: return launch<2>(a, cn_stream(stream));
    case 4: return launch<4>(a, cn_stream(stream));
    case 8: return launch<8>(a, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
}
