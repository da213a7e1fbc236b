// Device helpers shared by the fused MLP forward (mlp_fwd.hip), dgrad (mlp_bwd.hip) and wgrad (wgrad.hip) kernels.
//
// Execution model of the forward / dgrad kernels (modelled lane-by-lane in tests/test_layout_model.py):
//   * ONE wave64 owns 32 points and walks them through the whole network, one wave per SIMD (up to 512 registers);
//     every layer is computed transposed, Out^T[N x 32] = A[N x K] . B[K x 32], with v_mfma_f32_32x32x2_f32;
//   * A = a weight panel P[K/8][Np][8] (common.hpp): lane (i = lane&31, hh = lane>>5) loads 16 bytes at
//     ((kg*Np + 32t + i)*8 + 4hh) and feeds its 4 floats to MFMAs j = 0..3 of K-group kg, so MFMA (kg, j) contracts
//     the two k's 8kg + j (half-wave 0) and 8kg + 4 + j (half-wave 1);
//   * D = C-layout: lane (m = lane&31, hh), register r of tile t <-> feature 32t + 8(r>>2) + 4hh + (r&3), point m —
//     which is exactly the pairing above with kg = 4t + (r>>2), j = r&3.  Hence the accumulators of one layer ARE the
//     B operand of the next: B(kg, j) = X[kg>>2][4(kg&3) + j].  Hidden activations never leave the register file
//     (no LDS tile, no transposes, no barriers); only gamma(x) / gamma(d) go through a small LDS tile.
//
// Facts the schedules below are built on (scripts/coissue_probe.hip, scripts/opcost*_probe.hip; MI355X):
//   * v_mfma_f32_32x32x2_f32 issues every 64-66 cycles back to back; VALU instructions of ANY wave on the SIMD do
//     NOT overlap with it (fp32 MFMA and the VALU share the datapath: 157 TFLOP/s either way), so VALU work is a
//     straight tax on the MFMA rate and a second wave per SIMD hides nothing but memory latency;
//   * a 16-byte-per-lane load or store addressed as SGPR base + loop-invariant VGPR offset (buffer instructions)
//     issues for free behind an MFMA; the same access through a per-lane 64-bit pointer bumped with
//     v_add_co/v_addc costs ~25 cycles of MFMA issue; ds_read_b128 with an immediate offset is free as well;
//   * left alone, the machine scheduler sinks loads towards their first use (shorter live ranges) and exposes an
//     L2 round trip per K-group — hence the sched_barriers: the order written below is the order that executes.
#pragma once
#include "common.hpp"
#include "timing.hpp"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_z(float a, float b) {   // C = inline constant 0: starts an accumulation
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z, 0, 0, 0);
}

template <int NTO>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NTO]) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// ---- buffer addressing ---------------------------------------------------------------------------------------------
// Raw buffer resource (stride 0): address = base + voffset + soffset + imm; accesses at or past `bytes` are dropped
// (stores) or return 0 (loads) — used to discard the rows of padding points instead of masking every store.
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00027000);
}
__device__ __forceinline__ f32x4 buf_load(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// one float at a wave-uniform index of a buffer (biases of the VALU heads): through the resource, not the raw pointer — a
// global load keeps the 64-bit base in a VGPR pair for the whole kernel
__device__ __forceinline__ float buf_load1(rsrc_t r, int index) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, index * 4, 0));
}
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// Tile-major row blocks (common.hpp): inside a workgroup's 32-point tile row, column c of point m sits at byte
// tm_col(c) + m*32.  tm_col is wave-uniform (scalar unit); the per-lane part m*32 (+ 16 for the upper half-wave's
// 4 columns of an octet) is one loop-invariant VGPR, set out of range for padding points so that the hardware drops
// their stores and returns 0 for their loads.
__device__ __forceinline__ int tm_col(int c) { return (c >> 3) * 1024 + (c & 7) * 4; }
constexpr int TM_OOB = 0x7ffffff0;

// ReLU sign-bit words of one layer for this lane: MD dwords (common.hpp: s_mask)
template <int MD>
__device__ __forceinline__ void store_bits(rsrc_t r, int voff, int soff, const unsigned (&b)[MD]) {
  if (MD == 4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{b[0], b[MD > 1 ? 1 : 0], b[MD > 2 ? 2 : 0], b[MD > 3 ? 3 : 0]},
                                                      r, voff, soff, 0);
  else if (MD == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{b[0], b[MD > 1 ? 1 : 0]}, r, voff, soff, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(b[0], r, voff, soff, 0);
}
template <int MD>
__device__ __forceinline__ void load_bits(rsrc_t r, int voff, int soff, unsigned (&b)[MD]) {
  if (MD == 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
#pragma unroll
    for (int i = 0; i < MD; ++i) b[i] = v[i < 4 ? i : 0];
  } else if (MD == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
#pragma unroll
    for (int i = 0; i < MD; ++i) b[i] = v[i < 2 ? i : 0];
  } else {
    b[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
  }
}

// ---- A operand -----------------------------------------------------------------------------------------------------
// The packed weight blob behind ONE buffer resource + this lane's byte offset inside a 32-row x 8-k tile piece.
// x[lane] + x[lane ^ 32] in every lane, without a lane index: v_permlane32_swap exchanges the upper half of one register
// with the lower half of the other, so swapping x with itself leaves (lo, lo) and (hi, hi) — their sum is lo + hi in all 64
// lanes, the same two addends as x + __shfl_xor(x, 32) (which costs the lane id in a long-lived register + a ds_bpermute).
__device__ __forceinline__ float half_sum(float x) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // (not __builtin_bit_cast: on a vector ELEMENT it reads element 0)
}

struct APanel {
  rsrc_t rs;
  int lane;     // ((lane & 31) * 8 + 4 * (lane >> 5)) * 4
};

// tiles 0..NTO-1 of K-group kg of the panel at float offset `poff` (rows per group NP): the scalar offset carries
// panel, group and tile (SALU adds are free); the one VGPR offset is this lane's position inside a tile piece.
// (Constants added to the VGPR offset are NOT folded into the instruction's immediate by this compiler: they become
// distinct long-lived VGPRs, which at 512 registers means spills reloaded inside the MFMA stream.)
template <int NTO>
__device__ __forceinline__ void a_load(f32x4 (&a)[NTO], const APanel& P, int poff, int NP, int kg) {
  const int s0 = (poff + kg * NP * 8) * 4;
#pragma unroll
  for (int t = 0; t < NTO; ++t) a[t] = buf_load(P.rs, P.lane, s0 + t * 1024);
}

// Both A register sets of a panel: a0 <- group 0, a1 <- group 1.  Two sets are refilled IN PLACE by the GEMMs below:
// set a0 holds the even K-groups, a1 the odd ones; tile t of a set is reloaded with group kg+2 right behind its last
// MFMA of group kg (row j = 3), i.e. 5*NTO-1 MFMAs (~2600 cycles at NTO = 8) before its next use.  The register-operand
// GEMMs are fully unrolled, so the refills past the panel's last group are simply not emitted (9 % of the weight
// stream from L2 otherwise); the LDS-operand GEMM's loop clamps them instead (re-read): branch-free, so the compiler's
// vmcnt bookkeeping stays exact.
template <int NTO>
__device__ __forceinline__ void a_prefetch(f32x4 (&a0)[NTO], f32x4 (&a1)[NTO], const APanel& P, int poff, int NP,
                                           int last) {
  a_load<NTO>(a0, P, poff, NP, 0);
  a_load<NTO>(a1, P, poff, NP, last < 1 ? last : 1);
}

// Three-set variant for the register-operand GEMMs (prefetch distance 9*NTO-1 MFMAs): set s holds the groups
// kg == s (mod 3).  The extra distance is for the training kernels: a load issued behind a stash/gradient store is
// only consumable once that store is acknowledged (vmcnt is in order), and 2600 cycles do not always cover it.
template <int NTO>
__device__ __forceinline__ void a_prefetch3(f32x4 (&A)[3][NTO], const APanel& P, int poff, int NP, int last) {
  a_load<NTO>(A[0], P, poff, NP, 0);
  a_load<NTO>(A[1], P, poff, NP, last < 1 ? last : 1);
  a_load<NTO>(A[2], P, poff, NP, last < 2 ? last : 2);
}

struct NoSide {
  __device__ __forceinline__ void operator()(int, int, int) const {}
};

// ---- GEMM with the B operand in registers ----------------------------------------------------------------------------
// Q[t] (+)= sum_k Panel[k-group][32t + i] * X[k][m] for K = 32*NTI features held in C-layout registers X (see top).
// Fully unrolled (register indices must be static): 16*NTI*NTO MFMAs of straight-line code.
//   INIT   the first MFMA of every tile starts from 0 instead of Q
//   BIAS   the panel carries one more group whose k=0 column is the bias: one extra MFMA per tile vs B = (1, 0)
//   side(kg, j, t) is called once behind every MFMA: the caller's slots for independent memory instructions
//                  (stash / gradient stores of X, which stays live) that then cost no issue time.
template <int NTI, int NTO, bool BIAS, bool INIT, class Side = NoSide>
__device__ __forceinline__ void gemm_reg(f32x16 (&Q)[NTO], const f32x16 (&X)[NTI], f32x4 (&a0)[NTO], f32x4 (&a1)[NTO],
                                         const APanel& P, int poff, int NP, int hh, Side side = Side()) {
  constexpr int KG = 4 * NTI;
  constexpr int last = BIAS ? KG : KG - 1;
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) {
    f32x4 (&a)[NTO] = (kg & 1) ? a1 : a0;
    const int sn = (poff + (kg + 2 < last ? kg + 2 : last) * NP * 8) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = X[kg >> 2][4 * (kg & 3) + j];
#pragma unroll
      for (int t = 0; t < NTO; ++t) {
        if (INIT && kg == 0 && j == 0) Q[t] = mfma_z(a[t][j], b);
        else Q[t] = mfma(a[t][j], b, Q[t]);
        side(kg, j, t);
        if (j == 3 && kg + 2 <= last) a[t] = buf_load(P.rs, P.lane, sn + t * 1024);   // (kg is a compile-time constant here)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (BIAS) {   // group KG (even) was refilled into a0 behind group KG-2
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) Q[t] = mfma(a0[t][0], one, Q[t]);
  }
}

template <int NTI, int NTO, bool BIAS, bool INIT, class Side = NoSide>
__device__ __forceinline__ void gemm_reg3(f32x16 (&Q)[NTO], const f32x16 (&X)[NTI], f32x4 (&A)[3][NTO], const APanel& P,
                                          int poff, int NP, int hh, Side side = Side()) {
  constexpr int KG = 4 * NTI;
  constexpr int last = BIAS ? KG : KG - 1;
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) {
    f32x4 (&a)[NTO] = A[kg % 3];
    const int sn = (poff + (kg + 3 < last ? kg + 3 : last) * NP * 8) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = X[kg >> 2][4 * (kg & 3) + j];
#pragma unroll
      for (int t = 0; t < NTO; ++t) {
        if (INIT && kg == 0 && j == 0) Q[t] = mfma_z(a[t][j], b);
        else Q[t] = mfma(a[t][j], b, Q[t]);
        side(kg, j, t);
        if (j == 3 && kg + 3 <= last) a[t] = buf_load(P.rs, P.lane, sn + t * 1024);   // no refill past the last group
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (BIAS) {   // group KG was refilled into set KG % 3 behind group KG-3
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) Q[t] = mfma(A[KG % 3][t][0], one, Q[t]);
  }
}

// ---- GEMM with the B operand in an LDS tile (encodings) ------------------------------------------------------------
// Tile = 32 points x 64 floats, 16-byte chunks XOR-swizzled with (m & 15): conflict-free b128 for the wave.
__device__ __forceinline__ int enc_off(int m, int c) { return m * 64 + ((c ^ (m & 15)) << 2); }

// Copy this lane's chunks of an encoding tile to its stash block (lane (m, hh) owns columns 8c + 4hh .. +3)
__device__ __forceinline__ void stash_tile(const float* T, int chp, rsrc_t srs, int svo, int col, int m, int hh) {
  for (int c = 0; c < chp / 8; ++c)
    buf_store(srs, svo, tm_col(col) + c * 1024, *reinterpret_cast<const f32x4*>(T + enc_off(m, 2 * c + hh)));
}

// Q[t] (+)= sum_k Panel[k-group][32t + i] * T[m][k] over KG (even, >= 4) groups of 8 k's read from tile T; the chunk
// of group kg+1 is read behind the first MFMA of group kg.  The first four groups are peeled (INIT, and so that the
// loop carries no special cases).
template <int NTO, bool BIAS, bool INIT>
__device__ __forceinline__ void gemm_lds(f32x16 (&Q)[NTO], f32x4 (&a0)[NTO], f32x4 (&a1)[NTO], const APanel& P,
                                         int poff, int NP, int KG, const float* T, int m, int hh) {
  const int last = BIAS ? KG : KG - 1;
  // chunk 2kg+hh of row m: enc_off(m, 2kg + hh) * 4 == lbase ^ (kg << 5) (the XOR swizzle commutes with the group
  // bits), so one v_xor per group addresses it and no per-group address registers stay live
  const int lbase = m * 256 + ((hh ^ (m & 15)) << 4);
  auto ldb = [&](int kg) __attribute__((always_inline)) -> f32x4 {
    int lb = lbase;
    asm volatile("" : "+v"(lb));   // rematerialise the v_xor at every use: the hoisted addresses would be spilled
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(T) + (lb ^ ((kg < KG ? kg : KG - 1) << 5)));
  };
  auto step = [&](f32x4 (&a)[NTO], const f32x4& b, f32x4& bn, int kg, bool first) __attribute__((always_inline)) {
    const int sn = (poff + (kg + 2 < last ? kg + 2 : last) * NP * 8) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NTO; ++t) {
        if (INIT && first && j == 0) Q[t] = mfma_z(a[t][j], b[j]);
        else Q[t] = mfma(a[t][j], b[j], Q[t]);
        if (j == 0 && t == 0) bn = ldb(kg + 1);
        if (j == 3) a[t] = buf_load(P.rs, P.lane, sn + t * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  f32x4 b0 = ldb(0), b1 = b0;
  __builtin_amdgcn_sched_barrier(0);
  step(a0, b0, b1, 0, true);
  step(a1, b1, b0, 1, false);
  step(a0, b0, b1, 2, false);
  step(a1, b1, b0, 3, false);
  for (int kg = 4; kg < KG; kg += 2) {
    step(a0, b0, b1, kg, false);
    step(a1, b1, b0, kg + 1, false);
  }
  if (BIAS) {
    const float one = hh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NTO; ++t) Q[t] = mfma(a0[t][0], one, Q[t]);
  }
}

// Scheduling fence for a register tile set: an empty volatile asm that "rewrites" X.  SelectionDAG orders volatile asm
// with the other side-effecting nodes (loads, stores, sched_barriers), so the MFMAs that produce X cannot sink below
// this point and the ones that consume it cannot rise above it — without it, two back-to-back GEMMs in one basic
// block get their MFMAs pooled at the second one while all the first one's operand loads stay live (spills).
template <int NT>
__device__ __forceinline__ void pin(f32x16 (&X)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(X[t]));
}

// ---- epilogues -------------------------------------------------------------------------------------------------------
// In-place ReLU of NT tiles; when BITS (training), also packs the sign bits (x > 0) into MD = ceil(NT/2) dwords per
// lane (common.hpp s_mask).  Inference: one v_max per element.  Training: packed fp32 math on register pairs,
// 1.5 instructions per element instead of 3 (v_cmp + v_addc + v_max):
//     flag = clamp(x * +inf)   -> 1.0 where x > 0, else 0.0 (0 * inf = NaN clamps to 0 under DX10_CLAMP)
//     x    = x * flag          -> ReLU (negative inputs become -0.0, which behaves as 0 everywhere downstream)
//     acc  = acc * 2 + flag    -> the pair's two bit streams, exact in fp32 for the 16 bits a dword half holds
// Word layout: dword d covers tiles 2d, 2d+1; its upper 16 bits are the even registers r = 0, 2, .. 14 of tile 2d then
// of tile 2d+1 (first element = MSB), its lower 16 bits the odd registers likewise.  A last dword holding a single
// tile keeps 8 + 8 bits, left-aligned.  mask_bits consumes the words MSB first in exactly that element order.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT, bool BITS>
__device__ __forceinline__ void relu_bits(f32x16 (&Q)[NT], unsigned (&bits)[(NT + 1) / 2]) {
  const f32x2 inf2 = {__builtin_inff(), __builtin_inff()}, two2 = {2.f, 2.f};
  if (!BITS) {   // (the packed form measured faster than 16 v_max per tile: register pairs move together)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x = {Q[t][r], Q[t][r + 1]}, flag;
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(flag) : "v"(x), "v"(inf2));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(flag));
        Q[t][r] = x[0];
        Q[t][r + 1] = x[1];
      }
    return;
  }
#pragma unroll
  for (int d = 0; d < (NT + 1) / 2; ++d) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int t = 2 * d; t < 2 * d + 2 && t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x = {Q[t][r], Q[t][r + 1]}, flag;
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(flag) : "v"(x), "v"(inf2));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(flag));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(two2), "v"(flag));
        Q[t][r] = x[0];
        Q[t][r + 1] = x[1];
      }
    const bool single = 2 * d + 1 >= NT;
    bits[d] = ((unsigned)acc[0] << (single ? 24 : 16)) | ((unsigned)acc[1] << (single ? 16 : 0));
  }
}

// Gradient mask: Q <- Q * [bit], consuming the words produced by relu_bits MSB first in the same element order
// (v_add_co shifts the next bit into vcc, v_cndmask applies it).
template <int NT>
__device__ __forceinline__ void mask_bits(f32x16 (&Q)[NT], unsigned (&bits)[(NT + 1) / 2]) {
#pragma unroll
  for (int d = 0; d < (NT + 1) / 2; ++d)
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int t = 2 * d; t < 2 * d + 2 && t < NT; ++t)
#pragma unroll
        for (int r = par; r < 16; r += 2)
          asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(bits[d]), "+v"(Q[t][r]) : : "vcc");
}

// 32-feature-tile stores of C-layout registers into a tile-major row block, as a `side` functor of the GEMMs:
// 4*NTI stores (tile i>>2, quad i&3), ONE PER K-GROUP (the GEMM has exactly 4*NTI groups).  Spreading matters:
// vmcnt retires in order, stores included, so the A-operand loads issued behind a burst of stores cannot be consumed
// before the whole burst is acknowledged (measured: 8 stores per group in the first 4 groups cost the forward 12 %).
// `voff` = m*32 + hh*16 (TM_OOB for padding points), `soff` = tm_col(first column of the block): store (tile tt, quad
// q) is octet 4tt+q of the block = 1 KiB contiguous for the wave.
template <int NTI, int NTO>
struct TileStores {
  const f32x16 (&X)[NTI];
  rsrc_t rs;
  int voff, soff;
  __device__ __forceinline__ void operator()(int kg, int j, int t) const {
    if (j != 1 || t != 0) return;
    const int tt = kg >> 2, q = kg & 3;
    buf_store(rs, voff, soff + (4 * tt + q) * 1024, f32x4{X[tt][4 * q], X[tt][4 * q + 1], X[tt][4 * q + 2], X[tt][4 * q + 3]});
  }
};

template <int NT>
__device__ __forceinline__ void store_tiles(const f32x16 (&X)[NT], rsrc_t rs, int voff, int soff) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      buf_store(rs, voff, soff + (4 * t + q) * 1024, f32x4{X[t][4 * q], X[t][4 * q + 1], X[t][4 * q + 2], X[t][4 * q + 3]});
}
