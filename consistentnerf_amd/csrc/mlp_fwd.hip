// Fused positional-encoding + NeRF MLP forward on gfx950 matrix cores.
// Replaces run_network (R:37-52), Embedder.embed (H:15-63) and NeRF.forward (H:107-130): the 90-wide
// encodings and every [M,256] activation of the reference never reach HBM (inference), or reach it
// exactly once as the training stash.
//
// Mapping.  One wave64 owns 32 points and walks them through the whole network.  Every layer is
// computed TRANSPOSED, Out^T[N x 32] = W[N x K] . H^T[K x 32], with v_mfma_f32_32x32x2_f32 (exact
// fp32, bit-equal to an fmaf chain):
//   A operand  = weights, lane (i = lane&31, hh = lane>>5) holds W[n0+i][k + hh'] — one 16-byte load of
//                the packed panel (common.hpp) feeds 4 consecutive MFMAs; a wave reads 1 KiB contiguous.
//   B operand  = activations of the wave's 32 points, lane (m = lane&31, hh) holds H^T[k][m]; read from a
//                per-wave LDS tile Hs[m][k] (16-byte chunks XOR-swizzled by m&15: conflict-free b128).
//   D (C-layout) lane (m, hh) holds rows n = 32t + 8(r>>2) + 4hh + (r&3): 4 consecutive n per float4,
//                so ReLU'd accumulators go back to Hs with ds_write_b128 and straight into the next layer.
// No workgroup barriers, no cross-wave traffic; weights (2.4 MB/net) stay L2-resident and are streamed by
// every wave; MFMA-bound by construction (593 920 MAC per point at D=8/W=256 incl. K padding).
#include "mlp_common.hpp"

namespace {

struct FwdArgs {
  NetGeom g;
  const float* packed;
  const float* pts;
  const float* rays;
  const float* dirs;
  const float* z;
  float* raw;
  float* stash;
  int64_t M, Mp;
  int S, rs;
};

// gamma(x) channels of one point into Hs[m][0..chp): hh=0 lanes write x and the sines, hh=1 the cosines
// and the zero padding.  Channel order H:24-45: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(..)].
template <int W>
__device__ __forceinline__ void encode(float* Hs, const float (&x)[3], int L, int ch, int chp, int m, int hh) {
  auto put = [&](int k, float v) { Hs[hs_off<W>(m, k >> 2) + (k & 3)] = v; };
  if (hh == 0) {
    put(0, x[0]); put(1, x[1]); put(2, x[2]);
  } else {
    for (int k = ch; k < chp; ++k) put(k, 0.f);
  }
  float f = 1.f;
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float arg = x[d] * f;
      put(3 + 6 * l + 3 * hh + d, hh ? cosf(arg) : sinf(arg));
    }
    f *= 2.f;
  }
}

template <int NT, bool VD>
__global__ __launch_bounds__(64) void mlp_fwd_k(FwdArgs a) {
  constexpr int W = NT * 32;
  constexpr int NTH = NT / 2 > 0 ? NT / 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float Hs[];
  const NetGeom& g = a.g;
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p = (int64_t)blockIdx.x * 32 + m;
  const bool valid = p < a.M;
  const int64_t pc = valid ? p : a.M - 1;
  const int64_t ray = pc / a.S;
  const float* pk = a.packed;

  float x[3];
  if (a.pts != nullptr) {
    x[0] = a.pts[pc * 3 + 0]; x[1] = a.pts[pc * 3 + 1]; x[2] = a.pts[pc * 3 + 2];
  } else {
    const float* r = a.rays + ray * a.rs;
    const float zz = a.z[pc];
    x[0] = r[0] + r[3] * zz; x[1] = r[1] + r[4] * zz; x[2] = r[2] + r[5] * zz;   // R:384 (no FMA contraction)
  }
  // training: the stash row of the tile's first point (point-major [Mp][s_rows]); every block of the tile is
  // copied out of LDS with coalesced 1 KiB stores right after it is parked
  const int64_t pbase = (int64_t)blockIdx.x * 32;
  float* const st = a.stash != nullptr ? a.stash + pbase * g.s_rows : nullptr;
  auto stash_tile = [&](int col, int ncols) {
    if (st != nullptr) tile_to_global<W>(Hs, st + col, g.s_rows, ncols, pbase, a.M, lane);
  };
  // Ordering rule for every layer: [loads the next GEMM needs first: bias, A group 0] are queued BEFORE the
  // stash stores of the block just parked (see load_a0), the GEMM after them.
  encode<W>(Hs, x, g.L, g.in_ch, g.in_chp, m, hh);
  __builtin_amdgcn_wave_barrier();

  f32x16 acc[NT];
  f32x16 accs[NT];   // gamma(x) part of the skip layer, computed while gamma(x) is still in LDS
  f32x4 a0[NT];
  load_a0<NT>(a0, pk + g.f_l0, m, hh);
  init_bias<NT>(acc, pk + g.b_trunk[0], hh);
  stash_tile(g.s_enc, g.in_chp);
  gemm_seg<W, NT>(acc, pk + g.f_l0, W, g.in_chp / 8, Hs, m, hh, a0);
  if (g.skip >= 0) {
    init_bias<NT>(accs, pk + g.b_trunk[g.skip + 1], hh);
    gemm_seg<W, NT>(accs, pk + g.f_skip, W, g.in_chp / 8, Hs, m, hh);
  }
  __builtin_amdgcn_wave_barrier();
  park<W, NT, true>(acc, Hs, m, hh);
  __builtin_amdgcn_wave_barrier();

  for (int l = 1; l < g.D; ++l) {
    load_a0<NT>(a0, pk + g.f_trunk[l], m, hh);
    if (l == g.skip + 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = accs[t];
    } else {
      init_bias<NT>(acc, pk + g.b_trunk[l], hh);
    }
    stash_tile(g.s_h[l - 1], W);
    gemm_seg<W, NT>(acc, pk + g.f_trunk[l], W, W / 8, Hs, m, hh, a0);
    __builtin_amdgcn_wave_barrier();
    park<W, NT, true>(acc, Hs, m, hh);
    __builtin_amdgcn_wave_barrier();
  }

  if (!VD) {
    // output_linear (H:127-128): out[c] = b[c] + sum_k Wo[c][k] h[k]; each half-wave sums half the chunks
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = 0.f;
    for (int i = 0; i < W / 8; ++i) {
      const int ck = 2 * i + hh;
      const f32x4 h = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, ck));
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < g.out_ch) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_out + (int64_t)c * W + 4 * ck);
          o[c] += h[0] * w[0] + h[1] * w[1] + h[2] * w[2] + h[3] * w[3];
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] += __shfl_xor(o[c], 32, 64);
    if (valid && hh == 0)
      for (int c = 0; c < g.out_ch; ++c) a.raw[p * g.out_ch + c] = o[c] + pk[g.b_out + c];
    stash_tile(g.s_h[g.D - 1], W);
    return;
  } else {
    // sigma head (alpha_linear, H:117) on the VALU while the trunk output is in LDS
    load_a0<NT>(a0, pk + g.f_feat, m, hh);
    init_bias<NT>(acc, pk + g.b_feat, hh);
    float sig = 0.f;
    for (int i = 0; i < W / 8; ++i) {
      const int ck = 2 * i + hh;
      const f32x4 h = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, ck));
      const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_alpha + 4 * ck);
      sig += h[0] * w[0] + h[1] * w[1] + h[2] * w[2] + h[3] * w[3];
    }
    sig += __shfl_xor(sig, 32, 64);
    sig += pk[g.b_alpha];
    float v[3];
    {
      const float* dsrc = a.dirs != nullptr ? a.dirs + ray * 3 : a.rays + ray * a.rs + (a.rs - 3);
      v[0] = dsrc[0]; v[1] = dsrc[1]; v[2] = dsrc[2];
    }
    stash_tile(g.s_h[g.D - 1], W);
    // feature_linear (H:118), no activation
    gemm_seg<W, NT>(acc, pk + g.f_feat, W, W / 8, Hs, m, hh, a0);
    __builtin_amdgcn_wave_barrier();
    // gamma(viewdir) overwrites the (now dead) trunk tile; its share of views_linears first
    encode<W>(Hs, v, g.Ld, g.dir_ch, g.dir_chp, m, hh);
    __builtin_amdgcn_wave_barrier();
    f32x16 accv[NTH];
    f32x4 av0[NTH];
    load_a0<NTH>(av0, pk + g.f_viewsd, m, hh);
    init_bias<NTH>(accv, pk + g.b_views, hh);
    stash_tile(g.s_denc, g.dir_chp);
    gemm_seg<W, NTH>(accv, pk + g.f_viewsd, g.Wh, g.dir_chp / 8, Hs, m, hh, av0);
    __builtin_amdgcn_wave_barrier();
    park<W, NT, false>(acc, Hs, m, hh);
    __builtin_amdgcn_wave_barrier();
    load_a0<NTH>(av0, pk + g.f_views, m, hh);
    stash_tile(g.s_feat, W);
    gemm_seg<W, NTH>(accv, pk + g.f_views, g.Wh, W / 8, Hs, m, hh, av0);
    __builtin_amdgcn_wave_barrier();
    park<W, NTH, true>(accv, Hs, m, hh);      // ReLU in registers (rgb head below); LDS copy only feeds the stash
    __builtin_amdgcn_wave_barrier();
    // rgb_linear (H:125) straight from the accumulators: lane holds n = 32t + 8q + 4hh + j
    float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(pk + g.v_rgb + (int64_t)c * g.Wh + 32 * t + 8 * q + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[c] += accv[t][4 * q + j] * w[j];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] += __shfl_xor(o[c], 32, 64);
    if (valid && hh == 0) {
      *reinterpret_cast<float4*>(a.raw + p * 4) =
          make_float4(o[0] + pk[g.b_rgb + 0], o[1] + pk[g.b_rgb + 1], o[2] + pk[g.b_rgb + 2], sig);
    }
    stash_tile(g.s_hv, g.Wh);
  }
}

template <int NT>
int launch(const FwdArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  const size_t lds = (size_t)NT * 32 * 32 * sizeof(float);
  if (a.g.viewdirs) hipLaunchKernelGGL((mlp_fwd_k<NT, true>), dim3(grid), dim3(64), lds, st, a);
  else hipLaunchKernelGGL((mlp_fwd_k<NT, false>), dim3(grid), dim3(64), lds, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

}  // namespace

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
  a.packed = packed; a.pts = pts; a.rays = rays; a.dirs = dirs; a.z = z; a.raw = raw; a.stash = stash;
  a.M = B * S; a.Mp = cn_round_up(a.M, 32); a.S = S; a.rs = ray_stride;
  switch (a.g.NT) {
    case 2: return launch<2>(a, cn_stream(stream));
    case 4: return launch<4>(a, cn_stream(stream));
    case 8: return launch<8>(a, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
}
