// Device helpers shared by the fused MLP forward (mlp_fwd.hip) and backward (mlp_bwd.hip) kernels.
// Data-layout contract (also modelled lane-by-lane in tests/test_layout_model.py):
//   * a workgroup of TWO wave64 owns 32 points; every layer is computed transposed, Out^T[N x 32] = A[N x K] .
//     B[K x 32] with v_mfma_f32_32x32x2_f32, wave w producing output tiles [w*NT/2, (w+1)*NT/2) — 4 such
//     workgroups per CU = 2 waves per SIMD (<= 256 registers each), so one wave's layer epilogue / operand
//     latencies hide under the other's MFMAs;
//   * A = a weight panel P[K/8][Np][8] (common.hpp): lane (i = lane&31, hh = lane>>5) loads 16 bytes at
//     ((kg*Np + 32t + i)*8 + 4hh) and feeds its 4 floats to 4 consecutive MFMAs;
//   * B = the workgroup's LDS tile Hs[m][k] (32 points x W), 16-byte chunks XOR-swizzled with (m&15);
//   * D = C-layout: lane (m, hh), register r <-> row 32t + 8(r>>2) + 4hh + (r&3), column m.
#pragma once
#include "common.hpp"

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LDS tile addressing: point row m, 16-byte chunk c of the K axis.
template <int W>
__device__ __forceinline__ int hs_off(int m, int c) { return m * W + ((c ^ (m & 15)) << 2); }

// ---- variant used by the one-wave-per-tile backward kernel (mlp_bwd.hip): caller-visible first group ------------
// First A-operand group of a panel.  Issued by the caller BEFORE it queues the epilogue stores of the previous
// layer: vmcnt retires in order (stores included), so loads queued behind 32 KiB of stash stores would make the
// first MFMA of the next layer wait for the HBM write acknowledgements (~2-4 us per layer, measured as +20 %).
template <int NTO>
__device__ __forceinline__ void load_a0(f32x4 (&a0)[NTO], const float* __restrict__ panel, int m, int hh) {
  const float* pa = panel + ((int64_t)m * 8 + 4 * hh);
#pragma unroll
  for (int t = 0; t < NTO; ++t) a0[t] = *reinterpret_cast<const f32x4*>(pa + (int64_t)t * 256);
}

// acc[t] += sum_k P[k-panel][32t+i] * Hs[m][k]  for KG groups of 8 k's; panel rows per group = NP; a0 = group 0
// (load_a0).
template <int W, int NTO>
__device__ __forceinline__ void gemm_seg_a0(f32x16 (&acc)[NTO], const float* __restrict__ panel, int NP, int KG,
                                         const float* Hs, int m, int hh, f32x4 (&a0)[NTO]) {
  const float* pa = panel + ((int64_t)m * 8 + 4 * hh);
  f32x4 a1[NTO];
  // KG is even (all contracted widths are padded to multiples of 16).  Two register sets ping-pong so the
  // loads of group kg+1 / kg+2 are in flight under the 4*NTO MFMAs (64 cycles each) of group kg / kg+1;
  // the last prefetch is clamped (re-reads a valid group) to keep the loop branch-free for vmcnt counting.
  for (int kg = 0; kg < KG; kg += 2) {
    const float* p1 = pa + (int64_t)(kg + 1) * NP * 8;
#pragma unroll
    for (int t = 0; t < NTO; ++t) a1[t] = *reinterpret_cast<const f32x4*>(p1 + (int64_t)t * 256);
    {
      const f32x4 b = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * kg + hh));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NTO; ++t) acc[t] = mfma(a0[t][j], b[j], acc[t]);
    }
    const int k2 = kg + 2 < KG ? kg + 2 : kg;
    const float* p2 = pa + (int64_t)k2 * NP * 8;
#pragma unroll
    for (int t = 0; t < NTO; ++t) a0[t] = *reinterpret_cast<const f32x4*>(p2 + (int64_t)t * 256);
    {
      const f32x4 b = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * (kg + 1) + hh));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NTO; ++t) acc[t] = mfma(a1[t][j], b[j], acc[t]);
    }
  }
}

// A-operand prefetch ring: 2 register sets of NTO 16-byte pieces.  The A operand comes from L2 (weights are
// streamed, never staged); ring_start() queues group 0 of a panel and is called BEFORE the layer-boundary
// barriers / epilogue so their latency overlaps it; gemm_run() keeps group kg+1 in flight under the 4*NTO MFMAs
// (64 cycles each) of group kg (a 4-deep ring measured no faster and spills at 2 waves/SIMD).  Loads past the
// panel are clamped (re-read the last group) so the loop is branch-free and the compiler counts vmcnt exactly.
template <int NTO>
struct Ring {
  f32x4 r0[NTO], r1[NTO];
};

template <int NTO>
__device__ __forceinline__ void ring_load(f32x4 (&r)[NTO], const float* pa, int64_t gstride, int kg, int last) {
  const float* pg = pa + (int64_t)(kg < last ? kg : last) * gstride;
#pragma unroll
  for (int t = 0; t < NTO; ++t) {
#if defined(CN_EXP) && (CN_EXP & 1)   // ablation: no A-operand loads
    (void)pg; const float q = (float)kg * 1e-3f; r[t] = f32x4{q, q + 1e-4f, q, q};
#else
    r[t] = *reinterpret_cast<const f32x4*>(pg + (int64_t)t * 256);
#endif
  }
}

// panel rows per group = NP; `last` = index of the last group of the panel (KG-1, or KG when it has a bias group)
template <int NTO>
__device__ __forceinline__ void ring_start(Ring<NTO>& R, const float* __restrict__ panel, int NP, int last, int m,
                                           int hh) {
  const float* pa = panel + ((int64_t)m * 8 + 4 * hh);
  ring_load<NTO>(R.r0, pa, (int64_t)NP * 8, 0, last);
}

// acc[t] += sum_k P[k-group][32t+i] * Hs[m][k]  for KG groups of 8 k's (KG even: every contracted width is padded
// to a multiple of 32), then, if BIAS, acc += bias via the panel's extra group against the constant B operand
// (1, 0): one more MFMA per tile instead of a bias vector in registers.
template <int W, int NTO, bool BIAS>
__device__ __forceinline__ void gemm_run(f32x16 (&acc)[NTO], Ring<NTO>& R, const float* __restrict__ panel, int NP,
                                         int KG, const float* Hs, int m, int hh) {
  const float* pa = panel + ((int64_t)m * 8 + 4 * hh);
  const int64_t gs = (int64_t)NP * 8;
  const int last = BIAS ? KG : KG - 1;
  auto fma4 = [&](f32x4 (&r)[NTO], int kg) {
#if defined(CN_EXP) && (CN_EXP & 2)   // ablation: no LDS B reads
    const float q = (float)(kg + hh) * 1e-3f; const f32x4 b = {q, q, q + 1e-4f, q};
#else
    const f32x4 b = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * kg + hh));
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NTO; ++t) acc[t] = mfma(r[t][j], b[j], acc[t]);
  };
  for (int kg = 0; kg < KG; kg += 2) {
    ring_load<NTO>(R.r1, pa, gs, kg + 1, last); fma4(R.r0, kg);
    ring_load<NTO>(R.r0, pa, gs, kg + 2, last); fma4(R.r1, kg + 1);
  }
  if (BIAS) {   // group KG sits in r0 (KG even): P[KG][n][0] = bias[n]
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) acc[t] = mfma(R.r0[t][0], one, acc[t]);
  }
}

template <int W, int NTO, bool BIAS>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTO], const float* __restrict__ panel, int NP, int KG,
                                         const float* Hs, int m, int hh) {
  Ring<NTO> R;
  ring_start<NTO>(R, panel, NP, BIAS ? KG : KG - 1, m, hh);
  gemm_run<W, NTO, BIAS>(acc, R, panel, NP, KG, Hs, m, hh);
}

template <int NTO>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NTO]) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// ReLU (optional) the accumulators of output tiles t0..t0+NTO-1 and park them in the workgroup's LDS tile (B
// operand of the next layer) and, when training, in the stash.  The stash is POINT-MAJOR, [Mp][s_rows]: everything
// the backward needs about one point is one contiguous row, a lane's 4 consecutive features are ONE 16-byte store
// (straight from the registers, immediate offsets off one per-lane pointer), and the wgrad kernel can DMA 32-point
// slabs straight into LDS.  `sp` = this lane's (stash row + 4*hh) or nullptr; `col` = first column of the block.
// Lanes past M (padding points) store zeros so the backward never has to mask them.
template <int W, int NTO, bool RELU>
__device__ __forceinline__ void park(f32x16 (&acc)[NTO], float* Hs, bool to_lds, int t0, int m, int hh,
                                     float* __restrict__ sp, int col, bool valid) {
  float* dst = sp != nullptr ? sp + col + 32 * t0 : nullptr;
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = acc[t][4 * q + j];
        if (RELU) x = x > 0.f ? x : 0.f;
        acc[t][4 * q + j] = x;
        v[j] = x;
      }
#if defined(CN_EXP) && (CN_EXP & 4)   // ablation: no LDS tile writes
      if (to_lds && v[0] == 12345.678f) *reinterpret_cast<f32x4*>(Hs + hs_off<W>(m, 8 * (t0 + t) + 2 * q + hh)) = v;
#else
      if (to_lds) *reinterpret_cast<f32x4*>(Hs + hs_off<W>(m, 8 * (t0 + t) + 2 * q + hh)) = v;
#endif
      if (dst != nullptr) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(dst + 32 * t + 8 * q) = valid ? v : z;
      }
    }
}
