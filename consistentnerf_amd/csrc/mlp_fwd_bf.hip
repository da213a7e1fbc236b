// OPT-IN bf16-plane forward of the fused encoding + MLP (never the default, never the headline benchmark): inference at 1 / 2 / 3
// planes, and — three planes only, TRAIN = true — the training forward of the opt-in "bf16x3" training arithmetic, which writes
// the same stash as mlp_fwd_k (second bench line).  The GEMMs of R:37-52 / H:107-130 on v_mfma_f32_32x32x16_bf16
// (16x the fp32 MFMA rate nominally) with the operands split into NP bf16 "planes" and fp32 accumulation:
//     x = x0 + x1 + x2 (each plane the bf16 rounding of what the previous ones left), likewise w
//     NP = 1  plain bf16:      w0 x0                                             rel. error per product ~2^-9
//     NP = 2  "bf16x2":        w0 x0 + w0 x1 + w1 x0                             ~2^-16
//     NP = 3  "bf16x3":        + w0 x2 + w1 x1 + w2 x0  (6 of the 9 cross terms) ~2^-23, i.e. fp32-like
// Positional encodings, biases, the sigma / rgb heads and the accumulators stay fp32.
//
// Mapping: as mlp_fwd.hip — one wave64 owns 32 points and walks them through the whole network, every layer computed
// transposed (Out^T = W . H^T), hidden activations never leave the register file.  The C layout of a 32x32 MFMA tile
// (lane = point m + 32 hh, register r <-> feature 8(r>>2) + 4hh + (r&3)) is turned into the bf16 B operand of the next
// layer 8 registers at a time: K-step s = (tile t = s>>1, half = s&1) contracts the 16 features 32t + 16 half + [0, 16),
// lane (m, hh) supplying registers r = 8 half + e, e = 0..7 — the weight panels are packed in exactly that k order
// (pack_bf_k), so the A operand is one 16-byte buffer load per lane, 1 KiB contiguous per wave.  The split into planes
// runs on the VALU one register pair at a time in the gaps between MFMAs (v_cvt_pk_bf16_f32 + shift/and + sub).
// Bound: with the weights streamed per wave from L2 the kernel needs 16 B/clk/wave at NP = 2, 3 — the L2->CU limit
// (64 B/clk/CU) — so it is weight-stream-bound before it is MFMA-bound; sharing the panels through LDS is the next step.
#include <stdlib.h>

#include "encode.hpp"
#include "mlp_common.hpp"
#include "raygen.hpp"

#include "mlp_bf_common.hpp"

#ifndef CN_TRAIN_PAIR
#define CN_TRAIN_PAIR false     // tile pairing in the TRAINING forward's register GEMMs: measured level (fine 4.94 vs 4.99 ms, coarse 1.65 vs 1.58) at 13 instead of 6 spilled registers
#endif

namespace {

// ---- weight packing ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bf16_rne(float x) {   // round-to-nearest-even bf16 of a finite float, as 16 bits
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

struct BfPackJob {
  const float* src;        // weight [N, ld]
  int ld, col0, N, K;      // source rows / columns used (K <= Kp)
  int NTM;                 // 32-row tiles per K-step in memory (>= N/32)
  int Kp, kind;            // contracted width padded to 16; kind 0: k order of an LDS tile (16 s + 8 hh + e),
                           //                                kind 1: k order of C-layout registers (see the header)
                           //                                kind 2: as 1 for the TRANSPOSED map (dgrad): panel row r = source
                           //                                        column col0 + r, contracted index k = source row
  int64_t dst;             // byte offset
};
struct BfCopyJob { const float* src; int n; int64_t dst; };
struct BfPackArgs { BfPackJob job[48]; BfCopyJob cp[24]; int njobs, ncopies, NP; unsigned char* out; };

__global__ void pack_bf_k(BfPackArgs a_by_value) {
  (void)a_by_value;   // job table read in place from the kernarg segment (a by-value copy indexed by blockIdx.y lives in scratch)
  const __attribute__((address_space(4))) BfPackArgs& a = *(const __attribute__((address_space(4))) BfPackArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  if ((int)blockIdx.y >= a.njobs) {
    const __attribute__((address_space(4))) BfCopyJob& c = a.cp[blockIdx.y - a.njobs];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += gridDim.x * blockDim.x)
      reinterpret_cast<float*>(a.out + c.dst)[i] = c.src[i];
    return;
  }
  const __attribute__((address_space(4))) BfPackJob& j = a.job[blockIdx.y];
  const int NTO = j.NTM, NP = a.NP;             // tiles per K-step in memory (rows >= N are zero)
  const int64_t n = (int64_t)(j.Kp / 16) * NTO * 512;           // (s, to, i, hh, e): all planes of one weight together
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 7), hh = (int)((idx >> 3) & 1), i = (int)((idx >> 4) & 31);
    const int to = (int)((idx >> 9) % NTO), s = (int)((idx >> 9) / NTO);
    const int k = j.kind == 0 ? 16 * s + 8 * hh + e : 32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * hh + (e & 3);
    const int row = 32 * to + i;
    float w = 0.f;
    if (k < j.K && row < j.N) w = j.kind == 2 ? j.src[(int64_t)k * j.ld + j.col0 + row] : j.src[(int64_t)row * j.ld + j.col0 + k];
    unsigned short* dst = reinterpret_cast<unsigned short*>(a.out + j.dst);
    for (int p = 0; p < NP; ++p) {
      const unsigned h = bf16_rne(w);
      // piece layout [hh][i][e]: lane (i, hh) = lane 32 hh + i reads its 16 bytes at lane * 16 — consecutive lanes, consecutive
      // 16-byte chunks (with [i][hh] the lanes of a half-wave sat 32 bytes apart: a 2-way LDS bank conflict on every ds_read_b128
      // of the ring, SQ_LDS_BANK_CONFLICT = 50 % of the active LDS cycles)
      dst[((((int64_t)(s * NTO + to) * NP + p) * 2 + hh) * 32 + i) * 8 + e] = (unsigned short)h;
      w = w - __uint_as_float(h << 16);                          // exact: what this plane left over
    }
  }
}

// ---- kernel ----------------------------------------------------------------------------------------------------------
struct BfArgs {
  NetGeom g;
  BfGeom b;
  const unsigned char* pk;
  const float* pts;
  const float* rays;
  const float* dirs;
  const float* z;
  float* raw;
  float* stash;     // training stash of cnerf_mlp_fwd (same layout: the fp32 dgrad / wgrad kernels consume it) or nullptr
  int64_t M;
  int S, rs;
  RayGenDev cam;
};

template <int NTO, int NP, int NTM>
__device__ __forceinline__ void a_fetch(u32x4 (&A)[NTO][NP], const BfPanel& P, int poff, int s) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int p = 0; p < NP; ++p)
      A[t][p] = __builtin_amdgcn_raw_buffer_load_b128(P.rs, P.lane, poff + ((s * NTM + t) * NP + p) * 1024, 0);
}

// accumulators <- bias (fp32): register r of tile t is feature 32t + 8(r>>2) + 4hh + (r&3)
template <int NP>
struct ASets { static constexpr int N = NP == 1 ? 4 : (NP == 2 ? 2 : 1); };

// Q[t] += Panel . relu?(X) for the K = 32 NTI features held in C-layout registers X.  A-operand register sets: set
// s % NSET holds K-step s and is refilled in place with K-step s + NSET behind its last use.
template <int NTI, int NTO, int NP, bool RELU, int NTM>
__device__ __forceinline__ void gemm_bf_reg(f32x16 (&Q)[NTO], const f32x16 (&X)[NTI], const BfPanel& P, int poff) {
  constexpr int KS = 2 * NTI, NSET = ASets<NP>::N;
  u32x4 A[NSET][NTO][NP];
#pragma unroll
  for (int s = 0; s < NSET && s < KS; ++s) a_fetch<NTO, NP, NTM>(A[s], P, poff, s);
  u32x4 bc[NP], bn[NP];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair<NP, RELU>(X[0][2 * q], X[0][2 * q + 1], bc, q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    // tile-major: the 1 / 3 / 6 cross terms of a tile back to back (the matrix pipe forwards the accumulator; ordering them
    // product-major — no two consecutive MFMAs on one accumulator — measured 15 % SLOWER: more operands live, a read burst)
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
      products<NP>(Q[t], A[s % NSET][t], bc);
      if (s + NSET < KS) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
          A[s % NSET][t][p] = __builtin_amdgcn_raw_buffer_load_b128(P.rs, P.lane, poff + (((s + NSET) * NTM + t) * NP + p) * 1024, 0);
      }
      // the next K-step's B planes, one register pair behind each of the first four tiles' MFMAs
      if (s + 1 < KS && t < 4) {
        const int sn = s + 1;
        split_pair<NP, RELU>(X[sn >> 1][8 * (sn & 1) + 2 * t], X[sn >> 1][8 * (sn & 1) + 2 * t + 1], bn, t);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NTO < 4 && s + 1 < KS) {
#pragma unroll
      for (int q = NTO; q < 4; ++q) {
        const int sn = s + 1;
        split_pair<NP, RELU>(X[sn >> 1][8 * (sn & 1) + 2 * q], X[sn >> 1][8 * (sn & 1) + 2 * q + 1], bn, q);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) bc[p] = bn[p];
  }
}

// Q[t] += Panel . T for KS (even) K-steps of 16 channels read from the fp32 encoding tile T (channels 16 s + 8 hh + e);
// two A-operand sets alternate so that the panel pieces of step s + 1 are in flight under the MFMAs of step s.
template <int NTO, int NP, int NTM>
__device__ __forceinline__ void gemm_bf_lds(f32x16 (&Q)[NTO], const float* T, int KS, const BfPanel& P, int poff, int m, int hh) {
  u32x4 A0[NTO][NP], A1[NTO][NP];
  auto step = [&](u32x4 (&A)[NTO][NP], int s) __attribute__((always_inline)) {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * s + 2 * hh));
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * s + 2 * hh + 1));
    u32x4 b[NP];
    split_pair<NP, false>(c0[0], c0[1], b, 0);
    split_pair<NP, false>(c0[2], c0[3], b, 1);
    split_pair<NP, false>(c1[0], c1[1], b, 2);
    split_pair<NP, false>(c1[2], c1[3], b, 3);
#pragma unroll
    for (int t = 0; t < NTO; ++t) products<NP>(Q[t], A[t], b);
  };
  a_fetch<NTO, NP, NTM>(A0, P, poff, 0);
  for (int s = 0; s < KS; s += 2) {
    a_fetch<NTO, NP, NTM>(A1, P, poff, s + 1);
    __builtin_amdgcn_sched_barrier(0);
    step(A0, s);
    __builtin_amdgcn_sched_barrier(0);
    a_fetch<NTO, NP, NTM>(A0, P, poff, s + 2 < KS ? s + 2 : s);   // (the last refill re-reads step s: branch-free, unused)
    __builtin_amdgcn_sched_barrier(0);
    step(A1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT, int NP>
__global__ __launch_bounds__(64) void mlp_fwd_bf_k(BfArgs args_by_value) {
  constexpr int W = NT * 32, NTH = NT / 2;
  (void)args_by_value;
  const CN_CONST BfArgs& a = *(const CN_CONST BfArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const CN_CONST NetGeom& g = a.g;
  const CN_CONST BfGeom& bg = a.b;
  __shared__ __attribute__((aligned(16))) float Tx[32 * 64];   // gamma(x)
  __shared__ __attribute__((aligned(16))) float Td[32 * 64];   // gamma(d) (32 columns used)
  const int lane = threadIdx.x, m = lane & 31, hh = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (int)(a.M - p0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
  const int64_t ray = pc / a.S;
  const BfPanel P{make_rsrc(a.pk, (unsigned)bg.total), lane * 16};

  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (a.pts != nullptr) {
    x[0] = a.pts[pc * 3 + 0]; x[1] = a.pts[pc * 3 + 1]; x[2] = a.pts[pc * 3 + 2];
  } else if (a.cam.on) {
    float o[3], d[3];
    cn_gen_ray(a.cam, a.cam.first + ray, o, d, v);
    const float zz = a.z[pc];
    x[0] = o[0] + d[0] * zz; x[1] = o[1] + d[1] * zz; x[2] = o[2] + d[2] * zz;
  } else {
    const float* r = a.rays + ray * a.rs;
    const float zz = a.z[pc];
    x[0] = r[0] + r[3] * zz; x[1] = r[1] + r[4] * zz; x[2] = r[2] + r[5] * zz;   // R:384 (no FMA contraction)
  }
  if (!(a.pts == nullptr && a.cam.on)) {
    const float* dsrc = a.dirs != nullptr ? a.dirs + ray * 3 : a.rays + ray * a.rs + (a.rs - 3);
    v[0] = dsrc[0]; v[1] = dsrc[1]; v[2] = dsrc[2];
  }
  encode(Tx, x, g.L, g.in_ch, g.in_chp, m, hh, nullptr);
  encode(Td, v, g.Ld, g.dir_ch, g.dir_chp, m, hh, nullptr);

  f32x16 X[NT], Y[NT];
  // layer 0: gamma(x) from LDS -> Y
  bias_init<NT>(Y, P, (int)bg.b_trunk[0], hh);
  gemm_bf_lds<NT, NP, NT>(Y, Tx, g.in_chp / 16, P, (int)bg.p_l0, m, hh);
  pin<NT>(Y);
  // trunk layers l = 1..D-1 and feature_linear (l = D): Out = bias + W . relu(In) (+ the gamma(x) segment of the skip layer)
  float sig = 0.f;
  auto layer = [&](f32x16 (&In)[NT], f32x16 (&Out)[NT], int l) __attribute__((always_inline)) {
    bias_init<NT>(Out, P, (int)(l < g.D ? bg.b_trunk[l] : bg.b_feat), hh);
    if (l == g.D) {
      // h_{D-1} = relu(In) in place; the sigma head (alpha_linear, H:117) reads it on the VALU in fp32
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) In[t][r] = fmaxf(In[t][r], 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = buf_load(P.rs, hh * 16, (int)bg.v_alpha + (32 * t + 8 * q) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) sig = __builtin_fmaf(In[t][4 * q + j], wq[j], sig);
        }
      sig += __shfl_xor(sig, 32, 64);
      sig += *reinterpret_cast<const float*>(a.pk + bg.b_alpha);
    }
    // (the feature layer's input is already rectified: max(x, 0) once more is the identity — one GEMM body serves both)
    gemm_bf_reg<NT, NT, NP, true, NT>(Out, In, P, (int)(l < g.D ? bg.p_trunk[l] : bg.p_feat));
    {
      if (l == g.skip + 1) {
        pin<NT>(Out);
        gemm_bf_lds<NT, NP, NT>(Out, Tx, g.in_chp / 16, P, (int)bg.p_skip, m, hh);
      }
    }
    pin<NT>(Out);
  };
  const int nl = g.D;
  for (int l = 1; l <= nl; l += 2) {
    layer(Y, X, l);
    if (l + 1 <= nl) layer(X, Y, l + 1);
  }
  if (nl & 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) Y[t] = X[t];
  }
  // views_linears (H:120-123) on cat([feature, gamma(d)]): Y (registers, no activation on the feature) + Td (LDS)
  f32x16 V[NTH];
  bias_init<NTH>(V, P, (int)bg.b_views, hh);
  gemm_bf_reg<NT, NTH, NP, false, NT>(V, Y, P, (int)bg.p_views);
  pin<NTH>(V);
  gemm_bf_lds<NTH, NP, NT>(V, Td, g.dir_chp / 16, P, (int)bg.p_viewsd, m, hh);
  // rgb_linear (H:125) on relu(V), fp32 on the VALU
  float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wq = buf_load(P.rs, hh * 16, (int)bg.v_rgb + (c * (W / 2) + 32 * t + 8 * q) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[c] = __builtin_fmaf(fmaxf(V[t][4 * q + j], 0.f), wq[j], o[c]);
      }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] += __shfl_xor(o[c], 32, 64);
  const float* brgb = reinterpret_cast<const float*>(a.pk + bg.b_rgb);
  const rsrc_t ors = make_rsrc(a.raw + p0 * 4, (unsigned)(nvalid * 16));
  if (hh == 0) buf_store(ors, m * 16, 0, f32x4{o[0] + brgb[0], o[1] + brgb[1], o[2] + brgb[2], sig});
}

template <int NT, int NP, bool TRAIN>
__global__ __launch_bounds__(256) void mlp_fwd_bfs_k(BfArgs args_by_value) {
  constexpr int W = NT * 32, NTH = NT / 2;
  constexpr int MD = (NT + 1) / 2, MDV = (NTH + 1) / 2;
  (void)args_by_value;
  const CN_CONST BfArgs& a = *(const CN_CONST BfArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const CN_CONST NetGeom& g = a.g;
  const CN_CONST BfGeom& bg = a.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [4 ring slots][4 waves x (Tx 8 KiB | Td 8 KiB)]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Tx = reinterpret_cast<float*>(lds_raw + 4 * Ring<NT, NP>::SLOT + w * 16384);
  float* Td = Tx + 32 * 64;
  // a wave past the last point still moves its share of the panels and meets the barriers: it computes on clamped points
  // and its output resource is empty
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + w) * 32;
  const int64_t p = p0 + m;
  const int nvalid = a.M - p0 < 32 ? (a.M - p0 > 0 ? (int)(a.M - p0) : 0) : 32;
  const int64_t pc = p < a.M ? p : a.M - 1;
  const int64_t ray = pc / a.S;
  const BfPanel P{make_rsrc(a.pk, (unsigned)bg.total), lane * 16};
  // training: this WAVE's stash tile row (32 points; tile-major, mlp_common.hpp).  Lanes of padding points — and whole waves
  // past the last point, whose resource is empty — address out of range: the hardware drops their stores.
  const rsrc_t srs = make_rsrc(TRAIN && nvalid > 0 ? a.stash + p0 * g.s_rows : nullptr,
                               TRAIN && nvalid > 0 ? (unsigned)(32 * g.s_rows * 4) : 0u);
  const int svo = p < a.M ? m * 32 + hh * 16 : TM_OOB;
  const int smo = p < a.M ? m * 32 + hh * MD * 4 : TM_OOB;
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
  // start the panel stream before anything else: K-steps 0 and 1 of layer 0
  R.dma((int)bg.p_l0, 0, 0);
  R.dma((int)bg.p_l0, 1, 1);

  float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  if (a.pts != nullptr) {
    x[0] = a.pts[pc * 3 + 0]; x[1] = a.pts[pc * 3 + 1]; x[2] = a.pts[pc * 3 + 2];
  } else if (a.cam.on) {
    float o[3], d[3];
    cn_gen_ray(a.cam, a.cam.first + ray, o, d, v);
    const float zz = a.z[pc];
    x[0] = o[0] + d[0] * zz; x[1] = o[1] + d[1] * zz; x[2] = o[2] + d[2] * zz;
  } else {
    const float* r = a.rays + ray * a.rs;
    const float zz = a.z[pc];
    x[0] = r[0] + r[3] * zz; x[1] = r[1] + r[4] * zz; x[2] = r[2] + r[5] * zz;   // R:384 (no FMA contraction)
  }
  if (!(a.pts == nullptr && a.cam.on)) {
    const float* dsrc = a.dirs != nullptr ? a.dirs + ray * 3 : a.rays + ray * a.rs + (a.rs - 3);
    v[0] = dsrc[0]; v[1] = dsrc[1]; v[2] = dsrc[2];
  }
  encode(Tx, x, g.L, g.in_ch, g.in_chp, m, hh, nullptr);
  encode(Td, v, g.Ld, g.dir_ch, g.dir_chp, m, hh, nullptr);
  if (TRAIN) {
    stash_tile(Tx, g.in_chp, srs, svo, g.s_enc, m, hh);
    stash_tile(Td, g.dir_chp, srs, svo, g.s_denc, m, hh);
  }

  f32x16 X[NT], Y[NT];
  unsigned bits[MD];
  bias_init<NT>(Y, P, (int)bg.b_trunk[0], hh);
  pin<NT>(Y);                                  // (the bias loads are older than nothing the ring waits for below)
  R.template publish<Ring<NT, NP>::PW>();      // K-step 0 of layer 0 has landed everywhere
  gemm_ring_lds<4, NT, NT, NP, !TRAIN>(Y, Tx, R, (int)bg.p_l0, (int)(g.D > 1 ? bg.p_trunk[1] : bg.p_feat), m, hh);
  pin<NT>(Y);
  float sig = 0.f;
  auto layer = [&](f32x16 (&In)[NT], f32x16 (&Out)[NT], int l) __attribute__((always_inline)) {
    if (TRAIN) {
      // h_{l-1} = relu(In) in place + its sign bits (what the dgrad kernel masks with), exactly as mlp_fwd_k does
      relu_bits<NT, true>(In, bits);
      store_bits<MD>(srs, smo, tm_col(g.s_mask + g.s_mb[l - 1]), bits);
    }
    if (l == g.D) {
      if (!TRAIN) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) In[t][r] = fmaxf(In[t][r], 0.f);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = buf_load(P.rs, hh * 16, (int)bg.v_alpha + (32 * t + 8 * q) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) sig = __builtin_fmaf(In[t][4 * q + j], wq[j], sig);
        }
      sig += __shfl_xor(sig, 32, 64);
      sig += *reinterpret_cast<const float*>(a.pk + bg.b_alpha);
    }
    bias_init<NT>(Out, P, (int)(l < g.D ? bg.b_trunk[l] : bg.b_feat), hh);
    pin<NT>(Out);      // the bias loads retire (compiler waitcnt) before the ring's own vmcnt bookkeeping resumes
    const bool skip = l == g.skip + 1;
    // what follows this layer's register GEMM in the stream: its own gamma(x) segment, the next layer, or the view branch
    const int nxt = skip ? (int)bg.p_skip : (l + 1 < g.D ? (int)bg.p_trunk[l + 1] : (l + 1 == g.D ? (int)bg.p_feat : (int)bg.p_views));
    if (TRAIN)   // (the input is rectified already; its tiles go out to the stash under this GEMM)
      gemm_ring_reg<NT, NT, NT, NP, false, StashStores<NT>, 2, false, CN_TRAIN_PAIR>(Out, In, R, (int)(l < g.D ? bg.p_trunk[l] : bg.p_feat), nxt,
                                                                             StashStores<NT>{In, srs, svo, tm_col(g.s_h[l - 1])});
    else
      gemm_ring_reg<NT, NT, NT, NP, true>(Out, In, R, (int)(l < g.D ? bg.p_trunk[l] : bg.p_feat), nxt);
    if (skip) {
      pin<NT>(Out);
      gemm_ring_lds<4, NT, NT, NP, !TRAIN>(Out, Tx, R, (int)bg.p_skip, (int)(l + 1 < g.D ? bg.p_trunk[l + 1] : bg.p_feat), m, hh);
    }
    pin<NT>(Out);
  };
  const int nl = g.D;
  for (int l = 1; l <= nl; l += 2) {
    layer(Y, X, l);
    if (l + 1 <= nl) layer(X, Y, l + 1);
  }
  if (nl & 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) Y[t] = X[t];
  }
  f32x16 V[NTH];
  bias_init<NTH>(V, P, (int)bg.b_views, hh);
  pin<NTH>(V);
  if (TRAIN)   // the feature tiles (no activation: feature_linear is linear, H:118) go out under the view GEMM
    gemm_ring_reg<NT, NTH, NT, NP, false, StashStores<NT>, 2, false, CN_TRAIN_PAIR>(V, Y, R, (int)bg.p_views, (int)bg.p_viewsd,
                                                                            StashStores<NT>{Y, srs, svo, tm_col(g.s_feat)});
  else
    gemm_ring_reg<NT, NTH, NT, NP, false>(V, Y, R, (int)bg.p_views, (int)bg.p_viewsd);
  pin<NTH>(V);
  gemm_ring_lds<2, NTH, NT, NP, !TRAIN>(V, Td, R, (int)bg.p_viewsd, -1, m, hh);
  if (TRAIN) {
    unsigned bv[MDV];
    relu_bits<NTH, true>(V, bv);
    store_bits<MDV>(srs, p < a.M ? m * 32 + hh * MDV * 4 : TM_OOB, tm_col(g.s_mask + g.s_mb[g.D]), bv);
    store_tiles<NTH>(V, srs, svo, tm_col(g.s_hv));
  }
  float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wq = buf_load(P.rs, hh * 16, (int)bg.v_rgb + (c * (W / 2) + 32 * t + 8 * q) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[c] = __builtin_fmaf(TRAIN ? V[t][4 * q + j] : fmaxf(V[t][4 * q + j], 0.f), wq[j], o[c]);
      }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] += __shfl_xor(o[c], 32, 64);
  const float* brgb = reinterpret_cast<const float*>(a.pk + bg.b_rgb);
  const rsrc_t ors = make_rsrc(a.raw + (nvalid > 0 ? p0 : 0) * 4, (unsigned)(nvalid * 16));
  if (hh == 0) buf_store(ors, m * 16, 0, f32x4{o[0] + brgb[0], o[1] + brgb[1], o[2] + brgb[2], sig});
}

template <int NT, int NP, bool TRAIN>
int launch_bfs(const BfArgs& a, hipStream_t st) {
  const size_t lds = (size_t)4 * Ring<NT, NP>::SLOT + 4 * 16384;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CNERF_E_NODEVICE;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_bfs_k<NT, NP, TRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return (int)hipGetLastError();
    attr_set[dev] = true;
  }
  if (TRAIN) {
    const int64_t Mp = cn_round_up(a.M, 32);
    if (Mp > a.M) {   // last tile row holds padding points: the kernel drops their stores, wgrad reads them
      hipError_t e = hipMemsetAsync(a.stash + (Mp - 32) * a.g.s_rows, 0, (size_t)32 * a.g.s_rows * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  hipLaunchKernelGGL((mlp_fwd_bfs_k<NT, NP, TRAIN>), dim3((unsigned)cn_div_up(a.M, 128)), dim3(256), lds, st, a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

template <int NT>
int launch_bf(const BfArgs& a, int NP, hipStream_t st) {
  // shared-panel kernel by default; the per-wave one on request (CNERF_BF_PERWAVE=1: A/B measurements) or when the
  // encodings are not the 64- / 32-channel tiles its unrolled K-steps assume
  const char* e = getenv("CNERF_BF_PERWAVE");
  if (a.stash != nullptr) {   // training: the shared-panel kernel at three planes (the fp32-equivalent arithmetic) only
    if (NP != 3 || a.g.in_chp != 64 || a.g.dir_chp != 32) return CNERF_E_UNSUPPORTED;
    return launch_bfs<NT, 3, true>(a, st);
  }
  if (!(e && e[0] == '1') && a.g.in_chp == 64 && a.g.dir_chp == 32) {
    switch (NP) {
      case 1: return launch_bfs<NT, 1, false>(a, st);
      case 2: return launch_bfs<NT, 2, false>(a, st);
      case 3: return launch_bfs<NT, 3, false>(a, st);
      default: return CNERF_E_ARG;
    }
  }
  const unsigned grid = (unsigned)cn_div_up(a.M, 32);
  switch (NP) {
    case 1: hipLaunchKernelGGL((mlp_fwd_bf_k<NT, 1>), dim3(grid), dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL((mlp_fwd_bf_k<NT, 2>), dim3(grid), dim3(64), 0, st, a); break;
    case 3: hipLaunchKernelGGL((mlp_fwd_bf_k<NT, 3>), dim3(grid), dim3(64), 0, st, a); break;
    default: return CNERF_E_ARG;
  }
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

}  // namespace

extern "C" int64_t cnerf_packed_bf_bytes(const cnerf_net* net, int planes) {
  NetGeom g;
  BfGeom b;
  if (cn_make_geom(net, &g) || make_bf_geom(g, planes, &b)) return -1;
  return b.total;
}

extern "C" int cnerf_pack_weights_bf(const cnerf_net* net, const cnerf_ptrs* params, int planes, void* packed_bf,
                                     void* stream) {
  NetGeom g;
  BfGeom b;
  int rc = cn_make_geom(net, &g);
  if (rc) return rc;
  if ((rc = make_bf_geom(g, planes, &b))) return rc;
  if (!params || !packed_bf) return CNERF_E_ARG;
  const int nt = cnerf_num_tensors(net);
  for (int i = 0; i < nt; ++i)
    if (!params->p[i]) return CNERF_E_ARG;
  BfPackArgs a;
  a.njobs = a.ncopies = 0; a.NP = planes; a.out = static_cast<unsigned char*>(packed_bf);
  const int W = g.W, Wh = g.Wh, D = g.D;
  bool overflow = false;
  auto panel = [&](const float* src, int ld, int col0, int N, int K, int Kp, int kind, int64_t dst) {
    if (a.njobs >= 48) { overflow = true; return; }
    a.job[a.njobs++] = BfPackJob{src, ld, col0, N, K, g.NT, Kp, kind, dst};
  };
  auto copy = [&](const float* src, int n, int64_t dst) { a.cp[a.ncopies++] = BfCopyJob{src, n, dst}; };
  auto Wt = [&](int l) { return params->p[2 * l]; };
  auto Bt = [&](int l) { return params->p[2 * l + 1]; };
  panel(Wt(0), g.in_ch, 0, W, g.in_ch, g.in_chp, 0, b.p_l0);
  copy(Bt(0), W, b.b_trunk[0]);
  for (int l = 1; l < D; ++l) {
    const bool sk = g.skip >= 0 && l == g.skip + 1;
    panel(Wt(l), sk ? W + g.in_ch : W, sk ? g.in_ch : 0, W, W, W, 1, b.p_trunk[l]);
    if (sk) panel(Wt(l), W + g.in_ch, 0, W, g.in_ch, g.in_chp, 0, b.p_skip);
    copy(Bt(l), W, b.b_trunk[l]);
  }
  const int base = 2 * D;
  panel(params->p[base + 2], W, 0, W, W, W, 1, b.p_feat);
  copy(params->p[base + 3], W, b.b_feat);
  panel(params->p[base + 0], W + g.dir_ch, 0, Wh, W, W, 1, b.p_views);
  panel(params->p[base + 0], W + g.dir_ch, W, Wh, g.dir_ch, g.dir_chp, 0, b.p_viewsd);
  copy(params->p[base + 1], Wh, b.b_views);
  copy(params->p[base + 4], W, b.v_alpha);
  copy(params->p[base + 5], 1, b.b_alpha);
  copy(params->p[base + 6], 3 * Wh, b.v_rgb);
  copy(params->p[base + 7], 3, b.b_rgb);
  if (planes == 3) {   // the transposed panels of the bf16x3 dgrad (mlp_bwd_bf.hip)
    for (int l = 1; l < D; ++l) {
      const bool sk = g.skip >= 0 && l == g.skip + 1;
      panel(Wt(l), sk ? W + g.in_ch : W, sk ? g.in_ch : 0, W, W, W, 2, b.pt_trunk[l]);
    }
    panel(params->p[base + 2], W, 0, W, W, W, 2, b.pt_feat);
    panel(params->p[base + 0], W + g.dir_ch, 0, W, Wh, Wh, 2, b.pt_views);
  }
  if (overflow || a.ncopies > 24) return CNERF_E_UNSUPPORTED;
  hipLaunchKernelGGL(pack_bf_k, dim3(64, a.njobs + a.ncopies), dim3(256), 0, cn_stream(stream), a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_mlp_fwd_bf(const cnerf_net* net, const void* packed_bf, int planes, const float* pts,
                                const float* rays, int ray_stride, const float* dirs, const float* z, int64_t B, int S,
                                float* raw, void* stream) {
  BfArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if ((rc = make_bf_geom(a.g, planes, &a.b))) return rc;
  if (!packed_bf || !raw || B < 0 || S <= 0) return CNERF_E_ARG;
  if (!pts && (!rays || !z || ray_stride < 8)) return CNERF_E_ARG;
  if (!dirs && (!rays || ray_stride < 11)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.pk = static_cast<const unsigned char*>(packed_bf);
  a.pts = pts; a.rays = rays; a.dirs = dirs; a.z = z; a.raw = raw; a.stash = nullptr;
  a.M = B * S; a.S = S; a.rs = ray_stride;
  a.cam = cn_no_raygen();
  switch (a.g.NT) {
    case 4: return launch_bf<4>(a, planes, cn_stream(stream));
    case 8: return launch_bf<8>(a, planes, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
}

extern "C" int cnerf_mlp_fwd_bf_train(const cnerf_net* net, const void* packed_bf, const float* pts, const float* rays,
                                      int ray_stride, const float* dirs, const float* z, int64_t B, int S, float* raw,
                                      float* stash, void* stream) {
  BfArgs a;
  int rc = cn_make_geom(net, &a.g);
  if (rc) return rc;
  if ((rc = make_bf_geom(a.g, 3, &a.b))) return rc;
  if (!packed_bf || !raw || !stash || B < 0 || S <= 0) return CNERF_E_ARG;
  if (!pts && (!rays || !z || ray_stride < 8)) return CNERF_E_ARG;
  if (!dirs && (!rays || ray_stride < 11)) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  a.pk = static_cast<const unsigned char*>(packed_bf);
  a.pts = pts; a.rays = rays; a.dirs = dirs; a.z = z; a.raw = raw; a.stash = stash;
  a.M = B * S; a.S = S; a.rs = ray_stride;
  a.cam = cn_no_raygen();
  switch (a.g.NT) {
    case 4: return launch_bf<4>(a, 3, cn_stream(stream));
    case 8: return launch_bf<8>(a, 3, cn_stream(stream));
  }
  return CNERF_E_UNSUPPORTED;
}
