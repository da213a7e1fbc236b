// Fused positional-encoding + NeRF MLP forward on gfx950 matrix cores.
// Replaces run_network (R:37-52), Embedder.embed (H:15-63) and NeRF.forward (H:107-130): the 90-wide encodings and
// every [M,256] activation of the reference never reach HBM (inference), or reach it exactly once as the training
// stash.
//
// Mapping (mlp_common.hpp has the layouts).  One wave64 owns 32 points and walks them through the whole network, one
// wave per SIMD.  Every layer is computed TRANSPOSED, Out^T[N x 32] = W[N x K] . H^T[K x 32], with
// v_mfma_f32_32x32x2_f32 (exact fp32): the A operand is a weight panel streamed from L2 through a buffer resource
// (one free 16-byte load behind an MFMA feeds 4 of them), the B operand is the PREVIOUS layer's accumulator registers
// — the C layout of one layer is the B layout of the next — so hidden activations never leave the register file: no
// LDS tile, no transposes, no barriers.  Two accumulator sets X / Y alternate as input and output; ReLU runs in place.
// gamma(x) and gamma(d) are generated into a 2 x 8 KiB LDS tile (B operand of layer 0, of the skip segment and of the
// view branch).  Training stores each activation tile to the stash from the registers while the NEXT layer's MFMAs
// run (the tile is that GEMM's B operand, it stays live), plus 1 bit per hidden unit (ReLU sign) for the backward.
// MFMA-bound by construction: 593 920 MAC per point at D=8/W=256 incl. K padding.
#include "mlp_common.hpp"
#include "encode.hpp"
#include "raygen.hpp"

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
  RayGenDev cam;      // cam.on: the rays are those of a camera, generated here (rays == nullptr)
  const int* live;    // device count of LIVE rays, or nullptr: tiles whose points all lie at or beyond live * S write zero raw
                      // outputs and retire (cnerf_mlp_fwd_live: a batch padded to a fixed capacity, row count known on the device only)
};

#define CN_CONST __attribute__((address_space(4)))

// LIVE: the batch is padded to a fixed capacity and its live row count is read from device memory (cnerf_mlp_fwd_live) — a
// separate instantiation, so that the code of the default kernels is untouched by the gate (the D=8/W=256 training kernel sits at
// the register limit: the gate in the same body cost it an eighth spilled dword).
template <int NT, bool VD, bool TRAIN, bool LIVE = false>
__global__ __launch_bounds__(64) void mlp_fwd_k(FwdArgs args_by_value) {
  // the argument block is read in place from the kernarg segment (scalar loads next to their use): taken by value, all
  // of it is fetched by the kernel prologue and stays live in SGPRs across the layer loop (31 SGPR spills at NT = 8)
#ifdef CN_FWD_BYVAL
  const FwdArgs& a = args_by_value;
#else
  (void)args_by_value;
  const CN_CONST FwdArgs& a = *(const CN_CONST FwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#endif
  constexpr int W = NT * 32;
  constexpr int NTH = NT / 2 > 0 ? NT / 2 : 1;   // view-branch tiles (W/2 wide)
  constexpr int MD = (NT + 1) / 2, MDV = (NTH + 1) / 2;
  __shared__ __attribute__((aligned(16))) float Tx[32 * 64];   // gamma(x)
  __shared__ __attribute__((aligned(16))) float Td[32 * 64];   // gamma(d) (32 columns used)
#ifdef CN_FWD_BYVAL
  const NetGeom& g = a.g;
#else
  const CN_CONST NetGeom& g = a.g;
#endif
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (int)(a.M - p0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
  const int64_t ray = pc / a.S;
  if (LIVE) {
    // rows [live, B) of the batch are padding (the count was produced on the device, e.g. by cnerf_ss_batch): live * S is a
    // multiple of 32 whenever S is (64 / 192 here), so a tile is live or dead as a whole.  A dead tile leaves zero raw outputs —
    // sigma = 0: compositing gives weight 0 to every sample of a padding ray — and no stash (its backward is gated the same way).
    const int64_t lp = (int64_t)a.live[0] * a.S;
    if (p0 >= lp) {
      if (hh == 0 && p < a.M) {
        const int rc = VD ? 4 : g.out_ch;
        for (int c = 0; c < rc; ++c) a.raw[p * rc + c] = 0.f;
      }
      return;
    }
  }
  CN_TINIT(1)

  const APanel AP{make_rsrc(a.packed, (unsigned)(g.total * 4)), (m * 8 + 4 * hh) * 4};
  // this workgroup's stash tile row (tile-major, mlp_common.hpp); lanes of padding points address out of range: their
  // stores are dropped (the launcher zero-fills the last tile row once, the wgrad DMA reads it)
  const rsrc_t srs = make_rsrc(TRAIN ? a.stash + p0 * g.s_rows : nullptr, TRAIN ? (unsigned)(32 * g.s_rows * 4) : 0u);
  const int svo = p < a.M ? m * 32 + hh * 16 : TM_OOB;
  const int smo = p < a.M ? m * 32 + hh * MD * 4 : TM_OOB;   // sign-bit words of this lane: + tm_col(s_mask + s_mb[l])

  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  const float* const pre = a.emb != nullptr ? a.emb + pc * (g.in_ch + g.dir_ch) : nullptr;
  if (pre == nullptr) {
    if (a.pts != nullptr) {
      x[0] = a.pts[pc * 3 + 0]; x[1] = a.pts[pc * 3 + 1]; x[2] = a.pts[pc * 3 + 2];
    } else if (a.cam.on) {   // rays of a camera: generated per point (a few dozen flops next to 600 k MACs)
      float o[3], d[3];
      cn_gen_ray(a.cam, a.cam.first + ray, o, d, v);
      const float zz = a.z[pc];
      x[0] = o[0] + d[0] * zz; x[1] = o[1] + d[1] * zz; x[2] = o[2] + d[2] * zz;   // R:384 (no FMA contraction)
    } else {
      const float* r = a.rays + ray * a.rs;
      const float zz = a.z[pc];
      x[0] = r[0] + r[3] * zz; x[1] = r[1] + r[4] * zz; x[2] = r[2] + r[5] * zz;   // R:384 (no FMA contraction)
    }
    if (VD && !(a.pts == nullptr && a.cam.on)) {
      const float* dsrc = a.dirs != nullptr ? a.dirs + ray * 3 : a.rays + ray * a.rs + (a.rs - 3);
      v[0] = dsrc[0]; v[1] = dsrc[1]; v[2] = dsrc[2];
    }
  }
  f32x4 A[3][NT];   // A-operand register sets
  a_prefetch<NT>(A[0], A[1], AP, (int)g.f_l0, W, g.in_chp / 8);
  encode(Tx, x, g.L, g.in_ch, g.in_chp, m, hh, pre);
  if (VD) encode(Td, v, g.Ld, g.dir_ch, g.dir_chp, m, hh, pre != nullptr ? pre + g.in_ch : nullptr);
  if (TRAIN) {
    stash_tile(Tx, g.in_chp, srs, svo, g.s_enc, m, hh);
    if (VD) stash_tile(Td, g.dir_chp, srs, svo, g.s_denc, m, hh);
  }
  CN_T(0)

  f32x16 X[NT], Y[NT];
  unsigned bits[MD];
  // layer 0 (gamma(x) from LDS) -> Y
  gemm_lds<NT, true, true>(Y, A[0], A[1], AP, (int)g.f_l0, W, g.in_chp / 8, Tx, m, hh);
  CN_T(2)
  // One W x W layer, l = 1..D-1 trunk, l = D feature_linear (VD): ReLU the input set in place (+ sign bits), queue
  // the panel, GEMM into the other set while the input tiles go out to the stash.  The skip layer adds the gamma(x)
  // segment (and its bias) from LDS; the sigma head (alpha_linear, H:117) reads the trunk output on the VALU.
  float sig = 0.f;
  auto layer = [&](f32x16 (&In)[NT], f32x16 (&Out)[NT], int l) __attribute__((always_inline)) {
    const int poff = (int)(l < g.D ? g.f_trunk[l] : g.f_feat);
    a_prefetch3<NT>(A, AP, poff, W, W / 8);
    relu_bits<NT, TRAIN>(In, bits);
    if (TRAIN) store_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[l - 1]), bits);
    if (VD && l == g.D) {
      // weight quads in flight 16 at a time before their FMAs: consumed one by one, every load would be a separate
      // exposed L2 round trip (all 4*NT at once spill: both accumulator sets are live here)
      constexpr int TB = NT < 4 ? NT : 4;
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += TB) {
        f32x4 wq[TB][4];
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) wq[t][q] = buf_load(AP.rs, hh * 16, (int)(g.v_alpha + 32 * (t0 + t) + 8 * q) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sig = __builtin_fmaf(In[t0 + t][4 * q + j], wq[t][q][j], sig);
        __builtin_amdgcn_sched_barrier(0);
      }
      sig = half_sum(sig);
      sig += buf_load1(AP.rs, (int)g.b_alpha);
    }
    CN_T(3)
    if (TRAIN)
      gemm_reg3<NT, NT, true, true>(Out, In, A, AP, poff, W, hh, TileStores<NT, NT>{In, srs, svo, tm_col(g.s_h[l - 1])});
    else
      gemm_reg3<NT, NT, true, true>(Out, In, A, AP, poff, W, hh);
    if (l == g.skip + 1) {
      a_prefetch<NT>(A[0], A[1], AP, (int)g.f_skip, W, g.in_chp / 8);
      gemm_lds<NT, true, false>(Out, A[0], A[1], AP, (int)g.f_skip, W, g.in_chp / 8, Tx, m, hh);
    }
    CN_T(2)
  };
  // layer 0 wrote Y; layers alternate Y -> X -> Y ...; an odd layer count ends in X, an even one in Y.  The view
  // branch reads Y and the no-viewdirs head X: D = 8 needs no copy either way.
  const int nl = VD ? g.D : g.D - 1;
  for (int l = 1; l <= nl; l += 2) {
    layer(Y, X, l);
    if (l + 1 <= nl) layer(X, Y, l + 1);
  }
  if (VD && (nl & 1)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) Y[t] = X[t];
  }
  if (!VD && !(nl & 1)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) X[t] = Y[t];
  }
  const rsrc_t ors = make_rsrc(a.raw + p0 * (VD ? 4 : g.out_ch), (unsigned)(nvalid * (VD ? 4 : g.out_ch) * 4));

  if (!VD) {
    // trunk output h_{D-1} = relu(X); output_linear (H:127-128) on the VALU: out[c] = b[c] + sum_n Wo[c][n] h[n],
    // each lane sums its own features
    relu_bits<NT, TRAIN>(X, bits);
    if (TRAIN) {
      store_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[g.D - 1]), bits);
      store_tiles<NT>(X, srs, svo, tm_col(g.s_h[g.D - 1]));
    }
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < g.out_ch) {
            const f32x4 wv = buf_load(AP.rs, hh * 16, (int)(g.v_out + (int64_t)c * W + 32 * t + 8 * q) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[c] = __builtin_fmaf(X[t][4 * q + j], wv[j], o[c]);
          }
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = half_sum(o[c]);
    if (hh == 0)
      for (int c = 0; c < g.out_ch; ++c)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[c] + buf_load1(AP.rs, (int)g.b_out + c)), ors,
                                              (m * g.out_ch + c) * 4, 0, 0);
    CN_T(4)
    CN_TEND
    return;
  } else {
    // views_linears (H:120-123) on cat([feature, gamma(d)]): Y (K = W, registers) + Td (K = dir_chp, LDS, + bias)
    f32x16 V[NTH];
    f32x4 v0[NTH], v1[NTH];
    a_prefetch<NTH>(v0, v1, AP, (int)g.f_views, g.Wh, W / 8 - 1);
    if (TRAIN)
      gemm_reg<NT, NTH, false, true>(V, Y, v0, v1, AP, (int)g.f_views, g.Wh, hh,
                                     TileStores<NT, NTH>{Y, srs, svo, tm_col(g.s_feat)});
    else
      gemm_reg<NT, NTH, false, true>(V, Y, v0, v1, AP, (int)g.f_views, g.Wh, hh);
    pin<NTH>(V);
    a_prefetch<NTH>(v0, v1, AP, (int)g.f_viewsd, g.Wh, g.dir_chp / 8);
    gemm_lds<NTH, true, false>(V, v0, v1, AP, (int)g.f_viewsd, g.Wh, g.dir_chp / 8, Td, m, hh);
    CN_T(2)
    unsigned bv[MDV];
    relu_bits<NTH, TRAIN>(V, bv);
    if (TRAIN) {
      store_bits<MDV>(srs, p < a.M ? m * 32 + hh * MDV * 4 : TM_OOB, tm_col(g.s_mask + g.s_mb[g.D]), bv);
      store_tiles<NTH>(V, srs, svo, tm_col(g.s_hv));
    }
    CN_T(3)
    // rgb_linear (H:125) straight from the registers: lane holds n = 32t + 8q + 4hh + j
    float o[3] = {0.f, 0.f, 0.f};
    {
      f32x4 wq[3][NTH][4];                              // all weight quads in flight first (see the sigma head)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < NTH; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            wq[c][t][q] = buf_load(AP.rs, hh * 16, (int)(g.v_rgb + (int64_t)c * g.Wh + 32 * t + 8 * q) * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < NTH; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[c] = __builtin_fmaf(V[t][4 * q + j], wq[c][t][q][j], o[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = half_sum(o[c]);
    if (hh == 0)
      buf_store(ors, m * 16, 0, f32x4{o[0] + buf_load1(AP.rs, (int)g.b_rgb + 0), o[1] + buf_load1(AP.rs, (int)g.b_rgb + 1),
                                      o[2] + buf_load1(AP.rs, (int)g.b_rgb + 2), sig});
    CN_T(4)
    CN_TEND
  }
}

template <int NT>
int launch(const FwdArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  if (a.stash != nullptr) {
    if (a.Mp > a.M) {   // last tile row holds padding points: the kernel drops their stores, wgrad reads them
      hipError_t e = hipMemsetAsync(a.stash + (a.Mp - 32) * a.g.s_rows, 0, (size_t)32 * a.g.s_rows * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
    if (a.live != nullptr) {
      if (a.g.viewdirs) hipLaunchKernelGGL((mlp_fwd_k<NT, true, true, true>), dim3(grid), dim3(64), 0, st, a);
      else hipLaunchKernelGGL((mlp_fwd_k<NT, false, true, true>), dim3(grid), dim3(64), 0, st, a);
    } else if (a.g.viewdirs) hipLaunchKernelGGL((mlp_fwd_k<NT, true, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((mlp_fwd_k<NT, false, true>), dim3(grid), dim3(64), 0, st, a);
  } else {
    if (a.live != nullptr) return CNERF_E_UNSUPPORTED;      // (the device-side row count is a training-step feature)
    if (a.g.viewdirs) hipLaunchKernelGGL((mlp_fwd_k<NT, true, false>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((mlp_fwd_k<NT, false, false>), dim3(grid), dim3(64), 0, st, a);
  }
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

int dispatch(const FwdArgs& a, void* stream) {
  switch (a.g.NT) {
    case 2: return launch<2>(a, cn_stream(stream));
    case 4: return launch<4>(a, cn_stream(stream));
    case 8: return launch<8>(a, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
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
  a.cam = cn_no_raygen();
  a.live = nullptr;
  return dispatch(a, stream);
}

// cnerf_mlp_fwd on a batch padded to a fixed capacity of B rays whose LIVE row count sits in device memory (include/cnerf.h)
extern "C" int cnerf_mlp_fwd_live(const cnerf_net* net, const float* packed, const float* rays, int ray_stride, const float* z,
                                  int64_t B, int S, float* raw, float* stash, const int32_t* live_rays, void* stream) {
  FwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !raw || !rays || !z || !stash || !live_rays || B < 0 || S <= 0 || S % 32 != 0 || ray_stride < 8) return CNERF_E_ARG;
  if (a.g.viewdirs && ray_stride < 11) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.packed = packed; a.pts = nullptr; a.rays = rays; a.dirs = nullptr; a.z = z; a.emb = nullptr; a.raw = raw;
  a.stash = stash;
  a.M = B * S; a.Mp = cn_round_up(a.M, 32); a.S = S; a.rs = ray_stride;
  a.cam = cn_no_raygen();
  a.live = live_rays;
  return dispatch(a, stream);
}

// the fused encoding + MLP on the rays of a camera generated in-kernel (inference); used by cnerf_render_fwd_cam
int cn_mlp_fwd_cam(const cnerf_net* net, const float* packed, const RayGenDev& cam, const float* z, int64_t B, int S,
                   float* raw, void* stream) {
  FwdArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if (!packed || !raw || !z || B < 0 || S <= 0 || (a.g.viewdirs && !cam.vd)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.packed = packed; a.pts = nullptr; a.rays = nullptr; a.dirs = nullptr; a.z = z; a.emb = nullptr; a.raw = raw;
  a.stash = nullptr;
  a.M = B * S; a.Mp = cn_round_up(a.M, 32); a.S = S; a.rs = 0;
  a.cam = cam;
  a.live = nullptr;
  return dispatch(a, stream);
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
  a.cam = cn_no_raygen();
  a.live = nullptr;
  return dispatch(a, stream);
}
