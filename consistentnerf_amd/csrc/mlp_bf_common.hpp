// Shared pieces of the bf16-plane MLP kernels (mlp_fwd_bf.hip: forward, inference + opt-in training; mlp_bwd_bf.hip: the opt-in
// bf16x3 dgrad): packed-buffer geometry, the plane split, the MFMA products, and the LDS ring through which the four waves of
// a workgroup share one panel stream.  See mlp_fwd_bf.hip for the arithmetic and the mapping.
#pragma once
#include "mlp_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct BfGeom {                 // byte offsets into the packed buffer of cnerf_pack_weights_bf
  int64_t p_l0, p_trunk[16], p_skip, p_feat, p_views, p_viewsd;   // bf16 panels [K/16][N/32][NP][2 half-waves][32 lanes][8]
  int64_t pt_trunk[16], pt_feat, pt_views;                        // TRANSPOSED panels of the bf16x3 dgrad (NP == 3 only; else -1)
  int64_t b_trunk[16], b_feat, b_views, b_alpha, b_rgb;           // fp32 biases
  int64_t v_alpha, v_rgb;                                         // fp32 head weights [W], [3][W/2]
  int64_t total;
};

static int make_bf_geom(const NetGeom& g, int NP, BfGeom* b) {
  if (NP < 1 || NP > 3 || !g.viewdirs || (g.NT != 4 && g.NT != 8) || g.in_chp % 16 || g.dir_chp % 16) return CNERF_E_UNSUPPORTED;
  int64_t off = 0;
  auto panel = [&](int K, int N) { const int64_t o = off; off += (int64_t)(K / 16) * (N / 32) * NP * 1024; return o; };
  auto vec = [&](int n) { const int64_t o = off; off += cn_round_up((int64_t)n * 4, 64); return o; };
  b->p_l0 = panel(g.in_chp, g.W);
  for (int l = 1; l < g.D; ++l) b->p_trunk[l] = panel(g.W, g.W);
  b->p_skip = g.skip >= 0 ? panel(g.in_chp, g.W) : -1;
  b->p_feat = panel(g.W, g.W);
  b->p_views = panel(g.W, g.W);            // N = W/2 real rows; every K-step is padded to NT tiles (uniform K-step size:
  b->p_viewsd = panel(g.dir_chp, g.W);     // the LDS ring of the shared-panel kernel moves whole K-steps)
  // dgrad (mlp_bwd_bf.hip), three planes only: W_l^T for l = 1..D-1 (the h columns of the skip layer), feature_linear^T,
  // and the feature columns of views_linears^T (K = W/2 contracted, W output rows)
  for (int l = 0; l < 16; ++l) b->pt_trunk[l] = -1;
  b->pt_feat = b->pt_views = -1;
  if (NP == 3) {
    for (int l = 1; l < g.D; ++l) b->pt_trunk[l] = panel(g.W, g.W);
    b->pt_feat = panel(g.W, g.W);
    b->pt_views = panel(g.Wh, g.W);
  }
  for (int l = 0; l < g.D; ++l) b->b_trunk[l] = vec(g.W);
  b->b_feat = vec(g.W); b->b_views = vec(g.Wh); b->b_alpha = vec(1); b->b_rgb = vec(3);
  b->v_alpha = vec(g.W); b->v_rgb = vec(3 * g.Wh);
  b->total = off;
  return CNERF_OK;
}

namespace {

#ifndef CN_CONST
#define CN_CONST __attribute__((address_space(4)))
#endif

// v_cvt_pk_bf16_f32 (RNE).  Through the compiler, NOT inline asm: a VALU write needs two wait states before an MFMA reads
// the register on gfx950, and the hazard recognizer only inserts them (s_nop 1) for instructions it knows to be VALU.  As
// asm the conversion could sit one instruction in front of the MFMA that consumes it: that made the shared-panel kernel
// wrong at W = 128 / one plane (tile 0 of every encoding GEMM).  scripts/isa_hazards.py checks the ISA for this pattern.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// One register pair -> one dword of every plane (element 2q in the low half, 2q+1 in the high half).
template <int NP, bool RELU>
__device__ __forceinline__ void split_pair(float x0, float x1, u32x4 (&b)[NP], int q) {
  if (RELU) {   // one v_max each (fmaxf also emits a canonicalising v_max per operand: every VALU slot counts here)
    asm("v_max_f32 %0, 0, %1" : "=v"(x0) : "v"(x0));
    asm("v_max_f32 %0, 0, %1" : "=v"(x1) : "v"(x1));
  }
  unsigned h = cvt_pk_bf16(x0, x1);
  b[0][q] = h;
#pragma unroll
  for (int p = 1; p < NP; ++p) {
    x0 = x0 - __uint_as_float(h << 16);
    x1 = x1 - __uint_as_float(h & 0xffff0000u);
    h = cvt_pk_bf16(x0, x1);
    b[p][q] = h;
  }
}

// The three-plane split of one register pair in five stages (1, 4, 1, 4, 1 VALU instructions), so that a schedule can place
// them one stage per MFMA: a wave issues in order, and the 6 cross-term MFMAs of a tile are a dependent chain (each waits ~32
// cycles for the previous one) — side instructions placed BETWEEN them issue in those waits for free, a burst placed behind the
// chain leaves the matrix pipe idle for its whole length (measured before: ~42 instead of 32 cycles per MFMA).
struct Split3 {
  float x0, x1;
  unsigned h;
};
template <bool RELU>
__device__ __forceinline__ void split3_s0(Split3& S, float a, float b, u32x4 (&pl)[3], int q) {
  if (RELU) {
    asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(a));
    asm("v_max_f32 %0, 0, %1" : "=v"(b) : "v"(b));
  }
  S.x0 = a; S.x1 = b;
  S.h = cvt_pk_bf16(a, b);
  pl[0][q] = S.h;
}
__device__ __forceinline__ void split3_residual(Split3& S) {          // stages 1 and 3: what the plane just taken left over
  S.x0 = S.x0 - __uint_as_float(S.h << 16);
  S.x1 = S.x1 - __uint_as_float(S.h & 0xffff0000u);
}
__device__ __forceinline__ void split3_plane(Split3& S, u32x4 (&pl)[3], int p, int q) {   // stages 2 and 4
  S.h = cvt_pk_bf16(S.x0, S.x1);
  pl[p][q] = S.h;
}

__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the NP (NP + 1) / 2 cross terms w_i x_j with i + j < NP, smallest first
template <int NP>
__device__ __forceinline__ void products(f32x16& q, const u32x4 (&a)[NP], const u32x4 (&b)[NP]) {
#pragma unroll
  for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
    for (int i = 0; i <= sum; ++i) q = mfma_bf(a[i], b[sum - i], q);
}
// the same, starting the accumulation (C = 0 for the first product)
template <int NP>
__device__ __forceinline__ void products_init(f32x16& q, const u32x4 (&a)[NP], const u32x4 (&b)[NP]) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
    for (int i = 0; i <= sum; ++i) q = mfma_bf(a[i], b[sum - i], (sum == NP - 1 && i == 0) ? z : q);
}

struct BfPanel {
  rsrc_t rs;
  int lane;      // lane * 16 = (32 hh + m) * 16: this lane's 16 bytes inside a 1 KiB piece (layout [hh][m][8 bf16])
};

template <int NTO>
__device__ __forceinline__ void bias_init(f32x16 (&Q)[NTO], const BfPanel& P, int boff, int hh) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = buf_load(P.rs, hh * 16, boff + (32 * t + 8 * q) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) Q[t][4 * q + j] = v[j];
    }
}

// ======================================================================================================================
// Shared-panel variant: the four waves of a workgroup (one per SIMD, 32 points each) consume the SAME weight stream, so
// it crosses L2 -> CU once per 128 points instead of once per 32: the per-wave kernel above saturates at ~9.3 B/clk/wave
// (37 B/clk/CU) of panel traffic whatever the plane count — it is L2-stream-bound, MFMA-busy 25 / 43 / 65 % at 1 / 2 / 3
// planes.  Here a K-step of a panel (NT x NP pieces of 1 KiB) is moved HBM/L2 -> LDS by LDS-DMA (`buffer_load ... lds`, each
// wave a quarter of the pieces) into a 4-slot ring two K-steps ahead of its use, published by ONE barrier per K-step, and
// every wave reads its A operands from the ring with ds_read_b128 (64 lanes x 16 B contiguous: conflict-free).
//   iteration s:  DMA(K-step s+2 -> slot (s+2)&3)   [that slot was read last in iteration s-2: two barriers ago]
//                 MFMAs of K-step s from slot s&3
//                 s_waitcnt vmcnt(own pieces of s+2 may stay in flight) ; barrier          -> K-step s+1 is published
// Every GEMM has a multiple of 4 K-steps (the last one excepted), so each starts at slot 0 and hands the ring over to the
// next panel (whose first two K-steps it prefetches) without draining it.
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma1k(const i32x4& rs, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rs),
                 "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

template <int NT, int NP>
struct Ring {
  static constexpr int PIECES = NT * NP;            // 1 KiB pieces per K-step
  static constexpr int SLOT = PIECES * 1024;        // bytes per ring slot
  static constexpr int PW = PIECES / 4;             // DMA instructions per wave and K-step
  i32x4 rs;                                         // the packed buffer behind a buffer resource (DMA source)
  unsigned lds0;                                    // LDS byte address of slot 0 (wave-uniform)
  const unsigned char* ring;                        // the same, as a pointer for the ds_reads
  int w;                                            // wave index in the workgroup (scalar)
  int lane16;                                       // lane * 16: the DMA copies a piece lane-linearly
  int rd16;                                         // lane * 16: this lane's A-operand bytes inside a piece (conflict-free b128)
  // this wave's quarter of K-step `s` of the panel at byte offset `poff` -> slot
  __device__ __forceinline__ void dma(int poff, int s, int slot) const {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int i = w + 4 * j;
      dma1k(rs, lds0 + (unsigned)(slot * SLOT + i * 1024), lane16, poff + (s * PIECES + i) * 1024);
    }
  }
  // piece j (0..PW-1) of this wave's quarter: dealt out one per tile by the register-operand GEMM (a VMEM instruction
  // holds the in-order wave until the address unit takes it: four in a row cost 16-22 % of the kernel, measured)
  __device__ __forceinline__ void dma_piece(int poff, int s, int slot, int j) const {
    const int i = w + 4 * j;
    dma1k(rs, lds0 + (unsigned)(slot * SLOT + i * 1024), lane16, poff + (s * PIECES + i) * 1024);
  }
  __device__ __forceinline__ u32x4 a(int slot, int t, int p) const {
    return *reinterpret_cast<const u32x4*>(ring + slot * SLOT + (t * NP + p) * 1024 + rd16);
  }
  template <int OUTSTANDING>
  __device__ __forceinline__ void publish() const {   // own DMA pieces older than the newest OUTSTANDING have landed
    static_assert(OUTSTANDING >= 0 && OUTSTANDING < 64, "vmcnt range");
    // nothing is scheduled across the publish: the ring's only writer is the DMA asm, which the machine scheduler does not
    // see as a store to the LDS the next K-step's ds_reads load from
    __builtin_amdgcn_sched_barrier(0);
#ifndef CN_ABL_R_NODMA
    __builtin_amdgcn_s_waitcnt(0x0f70 | (OUTSTANDING & 15) | ((OUTSTANDING >> 4) << 14));   // vmcnt only (gfx9 encoding)
#endif
#ifndef CN_ABL_R_NOBAR
    __syncthreads();
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
};

// Q[t] += Panel . relu?(X), panel K-steps through the ring.  On entry K-steps 0 and 1 are in flight / published (K-step 0
// published); on exit the same holds for the NEXT panel (poff_next; -1: none follows).
// Training: the input tiles X (already rectified, fp32) go out to the stash while they are the B operand — the two register
// quads of K-step s (tile s >> 1, quads 2 (s & 1), + 1) behind tiles 0 and 1 of that step: one 16-byte store per lane each, a
// wave writes 1 KiB contiguous (mlp_common.hpp TileStores, dealt out for the 16-feature K-steps of this kernel).
#ifndef CN_STASH_AUX
#define CN_STASH_AUX 0   // cache-policy bits of the stash / gradient stores (experiment knob: sc0 = 1, nt = 2, sc1 = 16)
#endif
struct NoStash {
  __device__ __forceinline__ void operator()(int, int) const {}
};
template <int NTI>
struct StashStores {
  const f32x16 (&X)[NTI];
  rsrc_t rs;
  int voff, soff;
  __device__ __forceinline__ void operator()(int s, int t) const {
    if (t > 1) return;       // (behind tiles 0 and 1: every GEMM has at least two output tiles)
    const int tt = s >> 1, q = 2 * (s & 1) + t;
    const f32x4 v = {X[tt][4 * q], X[tt][4 * q + 1], X[tt][4 * q + 2], X[tt][4 * q + 3]};
    #ifdef CN_ABL_R_OOBSTORE   // (timing only: the store is issued, its data read, and dropped by the bounds check)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff | 0x40000000, soff + (4 * tt + q) * 1024, CN_STASH_AUX);
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff + (4 * tt + q) * 1024, CN_STASH_AUX);
#endif
  }
};

// Three planes: every side instruction of a tile (the three LDS reads of the A operand two tiles ahead, the DMA piece, the five
// stages of one register pair's split, the stash / gradient store) sits in the gap behind ONE of the tile's six MFMAs.
template <int NTI, int NTO, int NT, bool RELU, class Side, int SIDE_OPS, bool INIT, bool PAIR>
__device__ __forceinline__ void gemm_ring_reg3(f32x16 (&Q)[NTO], const f32x16 (&X)[NTI], const Ring<NT, 3>& R, int poff,
                                               int poff_next, Side side) {
  constexpr int NP = 3, KS = 2 * NTI;
  static_assert(KS % 4 == 0, "ring slot continuity");
  static_assert(!PAIR || NTO % 2 == 0, "tiles are processed in pairs");
  static_assert(NTO >= 4, "the publish sits behind tile NTO / 2 - 1, the cross-step prefetch in the last tiles");
#ifdef CN_R_NOPAIR
  constexpr int TP = 1;
#else
  constexpr int TP = PAIR ? 2 : 1;
#endif
#ifndef CN_R_AHEAD
#define CN_R_AHEAD 2
#endif
  constexpr int NA = 4, AH = (TP == 1) ? CN_R_AHEAD : 2;     // tiles the A reads run ahead of the MFMAs (TP + AH <= NA + ... sets)
  static_assert(AH + TP <= NA + (TP == 1 ? 0 : 0) && AH >= 2, "register sets");
  constexpr int IA[6] = {0, 1, 2, 0, 1, 0}, IB[6] = {2, 1, 0, 1, 0, 0};   // cross terms w_i x_j, i + j < 3, smallest first
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // K-step s + 1 is published in the MIDDLE of K-step s (behind tile NTO/2 - 1), not at its end: the A operands of its first two
  // tiles are then read in the gaps of the LAST two tiles of K-step s, so that no K-step starts with the four waves' first LDS reads
  // queued behind one another right after a barrier (ablation: the LDS reads are the largest single cost of these kernels, -23 %
  // without them).  The DMA of K-step s + 2 still goes out during K-step s: its slot was last read in K-step s - 2.
  // vector-memory operations this wave has issued in a K-step when it reaches the publish: the DMA pieces dealt to tiles
  // [0, NTO/2) and the side stores (tiles 0 and 1) — the publish may leave exactly those outstanding.
  constexpr int MID = NTO / 2;
  constexpr int PIECES_BEFORE = [] { int n = 0; for (int j = 0; j < Ring<NT, NP>::PW; ++j) n += (j % NTO) < MID; return n; }();
  constexpr int OUT = PIECES_BEFORE + SIDE_OPS;
  u32x4 bc[NP], bn[NP];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair<NP, RELU>(X[0][2 * q], X[0][2 * q + 1], bc, q);
  u32x4 A[NA][NP];      // tile g = s NTO + t of the GEMM lives in set g % 4 (NTO is a multiple of 4): reads run two tiles ahead
#pragma unroll
  for (int u = 0; u < AH; ++u)
#pragma unroll
    for (int p = 0; p < NP; ++p) A[u][p] = R.a(0, u, p);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    // PAIR: two tiles at a time, their six-MFMA chains alternating (the matrix pipe forwards the accumulator of a dependent
    // chain at full rate — scripts/mfma_bf16_probe.hip — so this only buys scheduling slack: +5 % on the inference kernel).
    // Every side instruction sits in one of the gaps behind an MFMA.
#pragma unroll
    for (int t = 0; t < NTO; t += TP) {
      const int sn = s + 1;
      Split3 S[TP];
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int u = 0; u < TP; ++u) {
          const int tt = t + u;
          const bool do_split = s + 1 < KS && tt < 4;
          Q[tt] = mfma_bf(A[tt % NA][IA[k]], bc[IB[k]], (INIT && s == 0 && k == 0) ? zero : Q[tt]);
#ifndef CN_ABL_R_NOLDS      // (ablation builds, timings only: -DCN_ABL_R_NOLDS / NODMA / NOSPLIT / NOSIDE / NOBAR)
          if (k < 3) {
            if (tt + AH < NTO) A[(tt + AH) % NA][k] = R.a(s & 3, tt + AH, k);
            else if (s + 1 < KS) A[(tt + AH) % NA][k] = R.a((s + 1) & 3, tt + AH - NTO, k);   // (published behind tile MID - 1)
          }
#endif
#ifndef CN_ABL_R_NODMA
          if (k == 3) {
#else
          if (k == 3 && false) {
#endif
#pragma unroll
            for (int j = tt; j < Ring<NT, NP>::PW; j += NTO) {
              if (s + 2 < KS) R.dma_piece(poff, s + 2, (s + 2) & 3, j);
              else if (poff_next >= 0) R.dma_piece(poff_next, s + 2 - KS, (s + 2) & 3, j);
            }
          }
#ifndef CN_ABL_R_NOSPLIT
          if (do_split) {
#else
          if (do_split && false) {
#endif
            if (k == 0) split3_s0<RELU>(S[u], X[sn >> 1][8 * (sn & 1) + 2 * tt], X[sn >> 1][8 * (sn & 1) + 2 * tt + 1], bn, tt);
            if (k == 1 || k == 3) split3_residual(S[u]);
            if (k == 2) split3_plane(S[u], bn, 1, tt);
            if (k == 4) split3_plane(S[u], bn, 2, tt);
          }
#ifndef CN_ABL_R_NOSIDE
          // the side stores ride behind tiles 0 and 1.  (Behind the LAST two tiles — younger than this K-step's DMA pieces in the
          // in-order vmcnt queue, so that no publish ever waits for an HBM store acknowledgement — measured level on the dgrad
          // and 10 % slower on the training forward: -DCN_R_SIDE_LAST.)
#ifdef CN_R_SIDE_LAST
          if (k == 5 && tt >= NTO - 2) side(s, tt - (NTO - 2));
#else
          if (k == 5) side(s, tt);
#endif
#endif
          __builtin_amdgcn_sched_barrier(0);
          if (k == 5 && tt == MID - 1) {     // K-step s + 1 (or the next panel's K-step 0) is published here
#ifdef CN_ABL_R_WAITSLACK   // (timing only, results race: the publish leaves the previous K-step's stores and later pieces in flight)
            if (s + 2 < KS || poff_next >= 0) R.template publish<OUT + (SIDE_OPS ? CN_ABL_R_WAITSLACK : 0)>();
#else
            if (s + 2 < KS || poff_next >= 0) R.template publish<OUT>();
#endif
            else R.template publish<SIDE_OPS>();      // (no DMA was issued in this K-step: only the side stores may be in flight)
          }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) bc[p] = bn[p];
  }
}

template <int NTI, int NTO, int NT, int NP, bool RELU, class Side = NoStash, int SIDE_OPS = 0, bool INIT = false, bool PAIR = true>
__device__ __forceinline__ void gemm_ring_reg(f32x16 (&Q)[NTO], const f32x16 (&X)[NTI], const Ring<NT, NP>& R, int poff,
                                              int poff_next, Side side = Side()) {
  // SIDE_OPS = vector-memory instructions `side` issues per K-step: they sit in the in-order vmcnt queue between this step's
  // DMA pieces, so the publish may leave that many more operations outstanding
  if constexpr (NP == 3 && NTO >= 4) {       // (the two-tile view GEMM of a W = 128 network keeps the plain schedule below)
#ifndef CN_BF3_BURST
    gemm_ring_reg3<NTI, NTO, NT, RELU, Side, SIDE_OPS, INIT, PAIR>(Q, X, R, poff, poff_next, side);
    return;
#endif
  }
  constexpr int KS = 2 * NTI, PW = Ring<NT, NP>::PW + SIDE_OPS;
  static_assert(KS % 4 == 0, "ring slot continuity");
  u32x4 bc[NP], bn[NP];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair<NP, RELU>(X[0][2 * q], X[0][2 * q + 1], bc, q);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 A[3][NP];       // A operands run two tiles ahead of the MFMAs that consume them (LDS latency)
#pragma unroll
    for (int p = 0; p < NP; ++p) A[0][p] = R.a(s & 3, 0, p);
    if (NTO > 1) {
#pragma unroll
      for (int p = 0; p < NP; ++p) A[1][p] = R.a(s & 3, 1, p);
    }
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
      if (t + 2 < NTO) {
#pragma unroll
        for (int p = 0; p < NP; ++p) A[(t + 2) % 3][p] = R.a(s & 3, t + 2, p);
      }
      if (INIT && s == 0) products_init<NP>(Q[t], A[t % 3], bc);
      else products<NP>(Q[t], A[t % 3], bc);
      // the DMA of K-step s+2, one piece behind each tile's MFMAs (all of them in front would delay the first LDS reads)
#pragma unroll
      for (int j = t; j < Ring<NT, NP>::PW; j += NTO) {
        if (s + 2 < KS) R.dma_piece(poff, s + 2, (s + 2) & 3, j);
        else if (poff_next >= 0) R.dma_piece(poff_next, s + 2 - KS, (s + 2) & 3, j);
      }
      if (s + 1 < KS && t < 4) {
        const int sn = s + 1;
        split_pair<NP, RELU>(X[sn >> 1][8 * (sn & 1) + 2 * t], X[sn >> 1][8 * (sn & 1) + 2 * t + 1], bn, t);
      }
      side(s, t);
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
    if (s + 2 < KS) R.template publish<PW>();
    else if (poff_next >= 0) R.template publish<PW>();
    else R.template publish<0>();
  }
}

// the same with the B operand read from the fp32 encoding tile T (KS K-steps of 16 channels: 16 s + 8 hh + e)
// three planes: as gemm_ring_reg3 — the A operand two tiles ahead, the DMA pieces and the split of the NEXT K-step's encoding
// chunk one stage per MFMA gap (the first K-step's planes are produced in the open)
template <int KS, int NTO, int NT>
__device__ __forceinline__ void gemm_ring_lds3(f32x16 (&Q)[NTO], const float* T, const Ring<NT, 3>& R, int poff, int poff_next,
                                               int m, int hh) {
  constexpr int NP = 3, PW = Ring<NT, NP>::PW;
  static_assert(NTO % 2 == 0, "tiles are processed in pairs");
  constexpr int IA[6] = {0, 1, 2, 0, 1, 0}, IB[6] = {2, 1, 0, 1, 0, 0};
  u32x4 bc[NP], bn[NP];
  {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 2 * hh));
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 2 * hh + 1));
    split_pair<NP, false>(c0[0], c0[1], bc, 0);
    split_pair<NP, false>(c0[2], c0[3], bc, 1);
    split_pair<NP, false>(c1[0], c1[1], bc, 2);
    split_pair<NP, false>(c1[2], c1[3], bc, 3);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 A[4][NP];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < NP; ++p) A[u][p] = R.a(s & 3, u, p);
    float cn[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (s + 1 < KS) {
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * (s + 1) + 2 * hh));
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * (s + 1) + 2 * hh + 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) { cn[e] = c0[e]; cn[4 + e] = c1[e]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTO; t += 2) {
      Split3 S[2];
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int tt = t + u;
          const bool do_split = s + 1 < KS && tt < 4;
          Q[tt] = mfma_bf(A[tt % 4][IA[k]], bc[IB[k]], Q[tt]);
          if (k < 3 && tt + 2 < NTO) A[(tt + 2) % 4][k] = R.a(s & 3, tt + 2, k);
          if (k == 3) {
#pragma unroll
            for (int j = tt; j < PW; j += NTO) {
              if (s + 2 < KS) R.dma_piece(poff, s + 2, (s + 2) & 3, j);
              else if (poff_next >= 0) R.dma_piece(poff_next, s + 2 - KS, (s + 2) & 3, j);
            }
          }
          if (do_split) {
            if (k == 0) split3_s0<false>(S[u], cn[2 * tt], cn[2 * tt + 1], bn, tt);
            if (k == 1 || k == 3) split3_residual(S[u]);
            if (k == 2) split3_plane(S[u], bn, 1, tt);
            if (k == 4) split3_plane(S[u], bn, 2, tt);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (NTO < 4 && s + 1 < KS) {
#pragma unroll
      for (int q = NTO; q < 4; ++q) split_pair<NP, false>(cn[2 * q], cn[2 * q + 1], bn, q);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) bc[p] = bn[p];
    if (s + 2 < KS || poff_next >= 0) R.template publish<PW>();
    else R.template publish<0>();
  }
}

template <int KS, int NTO, int NT, int NP, bool STAGED = true>
__device__ __forceinline__ void gemm_ring_lds(f32x16 (&Q)[NTO], const float* T, const Ring<NT, NP>& R, int poff, int poff_next,
                                              int m, int hh) {
  if constexpr (NP == 3 && STAGED) {
#ifndef CN_BF3_BURST
    gemm_ring_lds3<KS, NTO, NT>(Q, T, R, poff, poff_next, m, hh);
    return;
#endif
  }
  constexpr int PW = Ring<NT, NP>::PW;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 2 < KS) R.dma(poff, s + 2, (s + 2) & 3);
    else if (poff_next >= 0) R.dma(poff_next, s + 2 - KS, (s + 2) & 3);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * s + 2 * hh));
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(T + enc_off(m, 4 * s + 2 * hh + 1));
    u32x4 b[NP];
    split_pair<NP, false>(c0[0], c0[1], b, 0);
    split_pair<NP, false>(c0[2], c0[3], b, 1);
    split_pair<NP, false>(c1[0], c1[1], b, 2);
    split_pair<NP, false>(c1[2], c1[3], b, 3);
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
      u32x4 A[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) A[p] = R.a(s & 3, t, p);
      products<NP>(Q[t], A, b);
    }
    if (s + 2 < KS || poff_next >= 0) R.template publish<PW>();
    else R.template publish<0>();
  }
}

}  // namespace
