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
  auto ldb = [&](int kg) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * (kg < KG ? kg : KG - 1) + hh));
  };
  auto fma4 = [&](f32x4 (&r)[NTO], const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NTO; ++t) acc[t] = mfma(r[t][j], b[j], acc[t]);
  };
  // KG is even (all contracted widths are padded to multiples of 16).  Two register sets ping-pong: the A loads
  // and the B read of group kg+1 are issued BEFORE the 4*NTO MFMAs (64 cycles each) of group kg.  The
  // sched_barriers pin that order — left alone, the machine scheduler sinks loads towards their first use
  // (shorter live ranges), which exposed an L2 round trip on every second group (measured: dgrad 99 TFLOP/s).
  // The last prefetch is clamped (re-reads a valid group) to keep the loop branch-free for vmcnt counting.
  f32x4 b0 = ldb(0);
  for (int kg = 0; kg < KG; kg += 2) {
    const float* p1 = pa + (int64_t)(kg + 1) * NP * 8;
#pragma unroll
    for (int t = 0; t < NTO; ++t) a1[t] = *reinterpret_cast<const f32x4*>(p1 + (int64_t)t * 256);
    const f32x4 b1 = ldb(kg + 1);
    __builtin_amdgcn_sched_barrier(0);
    fma4(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int k2 = kg + 2 < KG ? kg + 2 : kg;
    const float* p2 = pa + (int64_t)k2 * NP * 8;
#pragma unroll
    for (int t = 0; t < NTO; ++t) a0[t] = *reinterpret_cast<const f32x4*>(p2 + (int64_t)t * 256);
    b0 = ldb(kg + 2);
    __builtin_amdgcn_sched_barrier(0);
    fma4(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- hand-scheduled GEMM core (one wave owns the whole 32-point tile: dgrad, forward) -------------------------------
// Facts this schedule is built on (scripts/coissue_probe.hip, MI355X):
//   * v_mfma_f32_32x32x2_f32 issues every ~66 cycles back to back; VALU instructions of ANY wave on the SIMD do not
//     overlap with it (fp32 MFMA and the VALU share the datapath: 157 TFLOP/s either way), so VALU work is a
//     straight tax and memory instructions are the only thing that can hide under an MFMA;
//   * a 16-byte-per-lane global load takes ~16 cycles to issue, so 8 of them in a row stall the in-order wave for
//     two MFMA slots; one load behind each MFMA is free;
//   * the machine scheduler, left alone, sinks loads towards their first use (live ranges) and exposes an L2
//     round trip per K-group — hence the sched_barriers: the order below is the order that executes.
// Two A register sets, refilled IN PLACE: set `a0` holds the even K-groups, `a1` the odd ones; tile t of a set is
// reloaded with group kg+2 right after its last MFMA of group kg (row j=3), i.e. 5*NTO-1 MFMAs (~2600 cycles at
// NTO=8) before its next use.  Both sets are loaded by the caller (a_prefetch) BEFORE the previous layer's epilogue
// stores: vmcnt retires in order, stores included, and loads queued behind 32 KiB of stores would wait for the HBM
// write acknowledgements.  The B operand (this lane's 16-byte chunk of the LDS tile) is read one group ahead.
// Groups past `last` are clamped (re-read): branch-free, so the compiler's vmcnt bookkeeping stays exact.
// Buffer addressing.  A 16-byte-per-lane load whose address is an SGPR base + a loop-invariant 32-bit VGPR offset
// (+ SGPR soffset) issues for free behind an MFMA; the same load through a per-lane 64-bit pointer that is bumped
// with v_add_co/v_addc costs the wave ~25 cycles of MFMA issue (scripts/opcost_probe*.hip).  So the weight blob is
// addressed through ONE buffer resource, panels and K-groups through the scalar offset.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00027000);
}
__device__ __forceinline__ f32x4 buf_load(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// A-operand source: the packed weight blob + this lane's byte offset inside a 32-row x 8-k tile piece
struct APanel {
  rsrc_t rs;
  int lane;     // ((lane & 31) * 8 + 4 * (lane >> 5)) * 4
};

// tile t of K-group kg of the panel at float offset `poff` (rows per group NP): soffset carries panel, group and the
// upper tile bit; the VGPR offset (lane + (t & 3) KiB) is loop invariant.
template <int NTO>
__device__ __forceinline__ void a_load(f32x4 (&a)[NTO], const APanel& P, int poff, int NP, int kg) {
  const int s0 = (poff + kg * NP * 8) * 4;
#pragma unroll
  for (int t = 0; t < NTO; ++t) a[t] = buf_load(P.rs, P.lane + (t & 3) * 1024, s0 + (t >> 2) * 4096);
}

template <int NTO>
__device__ __forceinline__ void a_prefetch(f32x4 (&a0)[NTO], f32x4 (&a1)[NTO], const APanel& P, int poff, int NP,
                                           int last) {
  a_load<NTO>(a0, P, poff, NP, 0);
  a_load<NTO>(a1, P, poff, NP, last < 1 ? last : 1);
}

// acc[t] += sum_k P[k-group][32t+i] * Hs[m][k] over KG (even, >= 4) groups of 8 k's; panel rows per group = NP.
// If BIAS, the panel carries one more group whose k=0 column is the bias: one extra MFMA per tile against B = (1, 0).
// `side(i)`, i = 0 .. 4*NTO-1, is called once behind each MFMA of row j=1 of the first four groups: the caller's
// slot for 4*NTO independent memory instructions (dgrad: the stash rows of the next ReLU mask) that then cost no
// issue time.  The four groups are peeled so that `i` is a compile-time constant after unrolling.
struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};

template <int W, int NTO, bool BIAS, class Side = NoSide>
__device__ __forceinline__ void gemm_pipe(f32x16 (&acc)[NTO], f32x4 (&a0)[NTO], f32x4 (&a1)[NTO], const APanel& P,
                                          int poff, int NP, int KG, const float* Hs, int m, int hh,
                                          Side side = Side()) {
  const int last = BIAS ? KG : KG - 1;
  auto ldb = [&](int kg) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * (kg < KG ? kg : KG - 1) + hh));
  };
  // one K-group: 4*NTO MFMAs from set `a` / chunk `b`; reads the next chunk `bn` early, refills `a` late
  auto step = [&](f32x4 (&a)[NTO], const f32x4& b, f32x4& bn, int kg, int sidx) {
    const int sn = (poff + (kg + 2 < last ? kg + 2 : last) * NP * 8) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NTO; ++t) {
        acc[t] = mfma(a[t][j], b[j], acc[t]);
#if !(defined(CN_EXP) && (CN_EXP & 2))    // ablation 2: no LDS B reads
        if (j == 0 && t == 0) bn = ldb(kg + 1);
#endif
#if !(defined(CN_EXP) && (CN_EXP & 16))   // ablation 16: no side loads
        if (j == 1 && sidx >= 0) side(sidx * NTO + t);
#endif
#if !(defined(CN_EXP) && (CN_EXP & 1))    // ablation 1: no A-operand refills
        if (j == 3) a[t] = buf_load(P.rs, P.lane + (t & 3) * 1024, sn + (t >> 2) * 4096);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  f32x4 b0 = ldb(0), b1 = b0;
  __builtin_amdgcn_sched_barrier(0);
  step(a0, b0, b1, 0, 0);
  step(a1, b1, b0, 1, 1);
  step(a0, b0, b1, 2, 2);
  step(a1, b1, b0, 3, 3);
  for (int kg = 4; kg < KG; kg += 2) {
    step(a0, b0, b1, kg, -1);
    step(a1, b1, b0, kg + 1, -1);
  }
  if (BIAS) {   // group KG (even) was refilled into a0 by the step of group KG-2
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) acc[t] = mfma(a0[t][0], one, acc[t]);
  }
}

// A-operand prefetch ring: CN_RING register sets of NTO 16-byte pieces.  The A operand comes from L2 (weights are
// streamed, never staged); ring_start() queues group 0 of a panel and is called BEFORE the layer-boundary
// barriers / epilogue so their latency overlaps it; gemm_run() keeps group kg+1 in flight under the 4*NTO MFMAs
// (64 cycles each) of group kg (a 4-deep ring measured no faster and spills at 2 waves/SIMD).  Loads past the
// panel are clamped (re-read the last group) so the loop is branch-free and the compiler counts vmcnt exactly.
#ifndef CN_RING
#define CN_RING 2
#endif
template <int NTO>
struct Ring {
  f32x4 r[CN_RING][NTO];
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
#pragma unroll
  for (int d = 0; d < CN_RING - 1; ++d) ring_load<NTO>(R.r[d], pa, (int64_t)NP * 8, d, last);
}

// acc[t] += sum_k P[k-group][32t+i] * Hs[m][k]  for KG groups of 8 k's (KG a multiple of CN_RING: every contracted
// width is padded to a multiple of 32), then, if BIAS, acc += bias via the panel's extra group against the constant
// B operand (1, 0): one more MFMA per tile instead of a bias vector in registers.
template <int W, int NTO, bool BIAS>
__device__ __forceinline__ void gemm_run(f32x16 (&acc)[NTO], Ring<NTO>& R, const float* __restrict__ panel, int NP,
                                         int KG, const float* Hs, int m, int hh) {
  const float* pa = panel + ((int64_t)m * 8 + 4 * hh);
  const int64_t gs = (int64_t)NP * 8;
  const int last = BIAS ? KG : KG - 1;
  auto fma4 = [&](f32x4 (&r)[NTO], int kg) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(Hs + hs_off<W>(m, 2 * kg + hh));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NTO; ++t) acc[t] = mfma(r[t][j], b[j], acc[t]);
  };
  // (Deliberately NOT pinned with sched_barriers: this two-waves-per-SIMD kernel measured slower with a dense MFMA
  // stream — fp32 MFMA and VALU share the SIMD's datapath, so the partner wave's VALU phases only progress in
  // the bubbles this loop leaves.  The one-wave kernels use gemm_pipe above.)
  for (int kg = 0; kg < KG; kg += CN_RING) {
#pragma unroll
    for (int d = 0; d < CN_RING; ++d) {
      ring_load<NTO>(R.r[(d + CN_RING - 1) % CN_RING], pa, gs, kg + d + CN_RING - 1, last);
      fma4(R.r[d], kg + d);
    }
  }
  if (BIAS) {   // group KG sits in r[0] (KG % CN_RING == 0): P[KG][n][0] = bias[n]
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) acc[t] = mfma(R.r[0][t][0], one, acc[t]);
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
#if defined(CN_NT)
        __builtin_nontemporal_store(valid ? v : z, reinterpret_cast<f32x4*>(dst + 32 * t + 8 * q));
#else
        *reinterpret_cast<f32x4*>(dst + 32 * t + 8 * q) = valid ? v : z;
#endif
      }
    }
}
