// Device helpers shared by the fused MLP forward (mlp_fwd.hip) and backward (mlp_bwd.hip) kernels.
// Data-layout contract (also modelled lane-by-lane in tests/test_layout_model.py):
//   * one wave64 owns 32 points; every layer is computed transposed, Out^T[N x 32] = A[N x K] . B[K x 32]
//     with v_mfma_f32_32x32x2_f32;
//   * A = a weight panel P[K/8][Np][8] (common.hpp): lane (i = lane&31, hh = lane>>5) loads 16 bytes at
//     ((kg*Np + 32t + i)*8 + 4hh) and feeds its 4 floats to 4 consecutive MFMAs;
//   * B = the wave's LDS tile Hs[m][k], 16-byte chunks XOR-swizzled with (m&15);
//   * D = C-layout: lane (m, hh), register r <-> row 32t + 8(r>>2) + 4hh + (r&3), column m.
#pragma once
#include "common.hpp"

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LDS tile addressing: point row m, 16-byte chunk c of the K axis.
template <int W>
__device__ __forceinline__ int hs_off(int m, int c) { return m * W + ((c ^ (m & 15)) << 2); }

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
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTO], const float* __restrict__ panel, int NP, int KG,
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

template <int W, int NTO>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTO], const float* __restrict__ panel, int NP, int KG,
                                         const float* Hs, int m, int hh) {
  f32x4 a0[NTO];
  load_a0<NTO>(a0, panel, m, hh);
  gemm_seg<W, NTO>(acc, panel, NP, KG, Hs, m, hh, a0);
}

template <int NTO>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NTO], const float* __restrict__ bias, int hh) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * q + 4 * hh);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = b[j];
    }
}

// ReLU (optional) the accumulators and park them in the wave's LDS tile (B operand of the next layer).
template <int W, int NTO, bool RELU>
__device__ __forceinline__ void park(f32x16 (&acc)[NTO], float* Hs, int m, int hh) {
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
      *reinterpret_cast<f32x4*>(Hs + hs_off<W>(m, 8 * t + 2 * q + hh)) = v;
    }
}

// Copy the first `ncols` (power of two, 16..W) columns of the wave's LDS tile to a POINT-MAJOR global block
// (training stash [Mp][s_rows] / gradient workspace [Mp][g_rows]): dst = row of the tile's first point + block
// column.  Everything the backward needs about one point is one contiguous row, so a tile column block is
// 32 x (ncols*4 B) contiguous pieces: each wave-instruction reads 64 x 16 B from LDS and stores 1 KiB fully
// coalesced (one point at ncols=256, two at 128, ...).  Padding points (>= M) are stored as ZEROS so the wgrad
// DMA never has to mask them.
template <int W>
__device__ __forceinline__ void tile_to_global(const float* Hs, float* __restrict__ dst, int64_t stride, int ncols,
                                               int64_t pbase, int64_t M, int lane) {
  const int cpp = ncols >> 2;              // 16-byte chunks per point
  const int sub = lane / cpp, c = lane - sub * cpp;
  const int step = 64 / cpp;               // points per wave-instruction
  for (int mm = sub; mm < 32; mm += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(mm, c));
    if (pbase + mm >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dst + (int64_t)mm * stride + 4 * c) = v;
  }
}
