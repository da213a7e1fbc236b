// Weight gradients of the NeRF MLP: dW[n][k] = sum_m X[m][n] * Y[m][k]  (m = points)
// with X = gradient w.r.t. a layer's pre-activation (columns of the workspace G[Mp][g_rows]) and
// Y = that layer's input (columns of the stash[Mp][s_rows]; both stored tile-major, common.hpp).  GEMMs on v_mfma_f32_32x32x2_f32 whose
// contraction runs over up to ~10^6 points.
//
// A workgroup (4 waves, one per SIMD, <= 512 registers each) owns one GEMM's whole (<=256 x <=256) output and one of
// `nsplit` point ranges.  The 4 waves tile the output as a gn x gk grid chosen per GEMM so that all four have work
// (256x256 -> 2x2 waves of 4x4 MFMA tiles; 128x256 -> 2x4 tiles each; 256x64 -> 2x2 each; 4x256 -> 1x2 ...).
// Operands move HBM -> LDS by DMA (buffer_load ... lds, 16 bytes per lane, no VGPR staging, scalar addressing): the
// stash / gradient workspace are stored as 32-point x 8-column blocks (common.hpp), one instruction moves one block
// (1 KiB contiguous) into a padded slot of the LDS image; the MFMA A operand of step s is one `ds_read_b32` per tile
// (column 32x + lane&31 of point 2s+hh; B likewise from Ys) at an IMMEDIATE offset from one per-lane base,
// conflict-free.  The inner loop is VALU-free apart from the packed bias column sums: fp32 MFMA and the VALU share the
// SIMD's datapath (mlp_common.hpp), every VALU instruction is MFMA time lost.  Reads run one step ahead of the
// MFMAs that consume them, and every side instruction of a step (reads, DMA pieces and their scalar address
// arithmetic, bias adds) is dealt out one per MFMA: a wave issues ~1 instruction per 4 cycles, a bunch of 25 between
// two MFMAs would idle the matrix pipe (sched_barriers pin the order).  2 LDS buffers for a 256 x 256 GEMM, 3-4 for
// the narrow ones: the DMA of slab s+1 is issued in the first half of slab s (so that it has landed at the barrier)
// and runs under its 16 steps x (an x ak) MFMAs; one barrier per slab.  The epilogue writes the accumulators with
// scalar-addressed buffer stores.
// GEMMs are launched largest-first over many small point ranges so the tail of the grid is short (measured makespan =
// the packing bound at M = 786 432); split partials are reduced in a fixed order by a second kernel (bit-reproducible
// run to run).
#include <math.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "mlp_bf_common.hpp"

namespace {

constexpr int MAX_WG_JOBS = 48;
#ifndef CN_WG_WAVES
#define CN_WG_WAVES 4
#endif
constexpr int NWAVES = CN_WG_WAVES;         // waves per workgroup (1 per SIMD)
constexpr int TM = 32;                      // points per LDS slab
constexpr int OCTF = 264;                   // LDS pitch (floats) of one 32-point x 8-column block: 256 + 8 (banks)
constexpr int LDS_BYTES = 160 * 1024;       // all of a CU's LDS: one workgroup per CU
constexpr int DUMMY_BYTES = 1024;           // landing zone of the no-op DMA pieces (out-of-range reads write zeros)
#ifdef CN_WGRAD_DYN
constexpr int LDS_FLOATS = (LDS_BYTES - DUMMY_BYTES - 16) / 4;   // + one ticket word
#else
constexpr int LDS_FLOATS = (LDS_BYTES - DUMMY_BYTES) / 4;
#endif

struct WgJob {
  int net;            // which operand set (WgArgs::net) the job reads / writes: 0, or 1 for the second network of a pair
  int xcol, ycol;     // first column of X in G rows, of Y in stash rows
  int N, K;           // output rows [n_lo, N) x cols [0, K) of the slab product are valid
  int n_lo;           // (rows below n_lo belong to another GEMM that shares the X columns; keeps DMA 16-B aligned)
  int tensor;         // destination parameter tensor
  int ld, col0;       // its row stride and first column
  int bias_tensor;    // -1: none
  int gk, an, ak;     // wave grid: wave w -> (wn, wk) = (w / gk, w % gk) owns an x ak tiles of 32x32
  int nsplit, chunk;  // this GEMM's point ranges: `nsplit` workgroups of `chunk` points (multiple of 32) each
  int first;          // linear block id of its first range (the grid is 1-D, jobs back to back, longest workgroups first)
  int bf3;            // 1: the opt-in bf16x3 body (wgrad_body_bf3) runs this GEMM; 0: exact fp32 (wgrad_body)
};

// One network's operands.  A launch carries up to two (the coarse and the fine network of a training step, whose
// backward passes are independent once the forward is done): their GEMMs share ONE grid, so the small launch of the
// coarse level (14 jobs x 64 point ranges on 256 CUs) rides in the tail of the fine one instead of having its own.
struct WgNet {
  int64_t toff[CNERF_MAX_TENSORS];   // offset of each tensor in the partial (parameter-space) buffer
  const float* stash;
  const float* G;
  float* partials;
  int64_t Mp, pstride;               // pstride = floats per split slice
  int s_rows, g_rows;
  const int* live;                   // device count of LIVE rays or nullptr (cnerf_mlp_bwd_live): points at or beyond
  int live_mul;                      // (live - live_sub) * live_mul (a multiple of 32) are padding — the ranges are cut over the live points
  int live_sub;                      // rays in front of this network's arrays that are not part of the launch (first_ray of the _live calls)
};

struct WgArgs {
  WgJob job[MAX_WG_JOBS];
  WgNet net[2];
  int nj;
  int total;          // virtual blocks of the launch (CN_WGRAD_DYN)
  int* counter;       // device ticket counter, zero before the launch (CN_WGRAD_DYN)
};

// The kernel argument block is read IN PLACE through the constant address space (scalar loads from the kernarg
// segment).  Taken by value and indexed with blockIdx.y, the compiler copies the whole struct to scratch first (2.7 KB
// per lane, ~200 scratch instructions, and every scratch access in the slab loop drains vmcnt, i.e. waits for the DMA
// in flight).
#define CN_CONST __attribute__((address_space(4)))
typedef const CN_CONST WgArgs WgArgsC;
typedef const CN_CONST WgJob WgJobC;
typedef const CN_CONST WgNet WgNetC;

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA instruction: 64 lanes x 16 bytes from (resource + voff + soff) to LDS bytes [lds_addr, +1 KiB), lane
// linear.  Inline asm on purpose: through the builtin the compiler treats every later ds_read as possibly aliasing
// the DMA target and drains vmcnt inside the first MFMA step of each slab — i.e. it waits for the prefetch of the
// NEXT slab — which serialises the double buffering.  Completion is awaited explicitly (vmcnt(0)) before the
// barrier that publishes the slab.
// Cache policy of the DMA loads: `nt` (non-temporal).  The operands are streamed exactly once per GEMM (10+ GB per C2 step)
// while the split partials the workgroups write (0.1-0.3 GB) are read back by the reduction right after the launch: with the
// stream marked non-temporal the partials survive in the cache hierarchy — wgrad + reduction 1.216 -> 1.188 ms at the 512-ray C4
// shard, 2.31 -> 2.29 ms at 1024 rays, level at 4096 (`sc1` / `sc0 sc1`: no effect; profiles/r03_wgrad_dma_policy.txt).
// -DCN_DMA_POL=0 builds without it.
#ifndef CN_DMA_POL
#define CN_DMA_POL 1
#endif
#if CN_DMA_POL == 1
#define CN_DMA_POLICY " nt"
#elif CN_DMA_POL == 2
#define CN_DMA_POLICY " sc1"
#elif CN_DMA_POL == 3
#define CN_DMA_POLICY " sc0 sc1"
#elif CN_DMA_POL == 4
#define CN_DMA_POLICY " sc1 nt"
#else
#define CN_DMA_POLICY ""
#endif
__device__ __forceinline__ void dma16(const i32x4& rs, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" CN_DMA_POLICY " lds"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rs),
                 "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}
__device__ __forceinline__ i32x4 dma_rsrc(const float* base, unsigned bytes) {
  const unsigned long long ba = (unsigned long long)base;
  return i32x4{__builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu)),
               __builtin_amdgcn_readfirstlane((int)((ba >> 32) & 0xffff)), __builtin_amdgcn_readfirstlane((int)bytes),
               0x00027000};
}

// Body for a compile-time per-wave tile block AN x AK (<= 4 x 4); BS: this wave also sums the X columns (bias).
template <int AN, int AK, bool BS>
__device__ __forceinline__ void wgrad_body(WgNetC& a, WgJobC& jb, float* lds, const int vb) {
  const int split = __builtin_amdgcn_readfirstlane(vb - jb.first);
  const int tid = threadIdx.x, lane = tid & 63, i31 = lane & 31, hh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on the scalar unit: addresses stay SALU
  const int wn = wv / jb.gk, wk = wv - wn * jb.gk;
  // (the by-value argument struct is indexed dynamically, so it lives in scratch: anything the slab loop needs is
  // pulled into SGPRs here — a scratch_load inside the loop would also drain vmcnt, i.e. wait for the DMA in flight)
  const int g_rows4 = __builtin_amdgcn_readfirstlane(a.g_rows * 4), s_rows4 = __builtin_amdgcn_readfirstlane(a.s_rows * 4);
  int64_t Mp = a.Mp;
  int64_t chunk = jb.chunk;
  if (a.live != nullptr) {
    // the batch's live row count lives on the device (scalar load): the GEMM's `nsplit` point ranges are re-cut over the LIVE
    // points, so that every workgroup of the launch stays equally long (ranges planned for the capacity and clipped would leave the
    // last few per GEMM empty: 1708 instead of 1792 workgroups on 256 CUs = 7 rounds of time for 6.7 of work, measured -4 %)
    int64_t lp = ((int64_t)a.live[0] - a.live_sub) * a.live_mul;
    lp = lp < 0 ? 0 : lp;
    Mp = lp < Mp ? lp : Mp;
    const int64_t slabs = Mp / TM;
    chunk = ((slabs + jb.nsplit - 1) / jb.nsplit) * TM;
  }
  const int64_t m_begin0 = (int64_t)split * chunk;
  const int64_t m_begin = m_begin0 < Mp ? m_begin0 : Mp;      // (a range past the end is empty: zero-sized resources, zero partial)
  const int64_t m_end = m_begin + chunk < Mp ? m_begin + chunk : Mp;
  const int nslab = __builtin_amdgcn_readfirstlane(m_end > m_begin ? (int)((m_end - m_begin) / TM) : 0);
  const int ntn = (jb.N + 31) >> 5, ntk = (jb.K + 31) >> 5;
  const int tn0 = __builtin_amdgcn_readfirstlane(wn * AN), tk0 = __builtin_amdgcn_readfirstlane(wk * AK);   // first n / k tile of this wave
  const bool active = tn0 < ntn && tk0 < ntk;      // wave-uniform
  // this split's rows of the two operands behind buffer resources; a lane past the operand's width reads out of
  // range (zeros land in the unused columns of the LDS row)
  // this split's tile rows of the two operands (tile-major storage, common.hpp), from the operand's first column octet
  const i32x4 xr = dma_rsrc(a.G + m_begin * a.g_rows + (jb.xcol >> 3) * 256, (unsigned)((m_end - m_begin) * a.g_rows * 4));
  const i32x4 yr = dma_rsrc(a.stash + m_begin * a.s_rows + (jb.ycol >> 3) * 256, (unsigned)((m_end - m_begin) * a.s_rows * 4));
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds);
  const int vo = lane * 16;
  CN_TINIT(NWAVES)
  f32x16 acc[AN][AK];
#pragma unroll
  for (int x = 0; x < AN; ++x)
#pragma unroll
    for (int y = 0; y < AK; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  // bias column sums, two tiles per v_pk_add_f32 (every VALU instruction costs MFMA issue time: ~7 cycles each)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int AN2 = (AN + 1) / 2;
  f32x2 bsum2[AN2];
#pragma unroll
  for (int x = 0; x < AN2; ++x) bsum2[x] = f32x2{0.f, 0.f};

  // LDS = `nbuf` slab buffers of [X image | Y image], sized by this GEMM's operand widths: a full 256 x 256 GEMM
  // double-buffers (2 x 66 KiB); the narrow ones (heads, gamma columns) are latency-bound, not MFMA-bound, and get
  // 3-4 buffers so that 2-3 slabs are in flight.
  const int xoct = __builtin_amdgcn_readfirstlane(4 * ntn), yoct = __builtin_amdgcn_readfirstlane(4 * ntk);
  const int bufF = (xoct + yoct) * OCTF;
  int nbuf = LDS_FLOATS / bufF;
  nbuf = nbuf > 4 ? 4 : nbuf;
  // DMA of slab `sl` (= one 32-point tile row) into its buffer: one instruction moves one 32-point x 8-column block
  // (1 KiB contiguous in HBM) to its padded slot of the LDS image; blocks round-robin over the waves, 2 x 8
  // instructions per wave: piece i (0..15) is column octet wv + 4*(i>>1) of X (i even) or Y (i odd).  Octets past the
  // operand's width and slabs past the end read out of range; their zeros land in a 1 KiB dummy block at the end of LDS
  // (they stay in the instruction stream so that the vmcnt bookkeeping of `publish` is a constant).
  auto piece = [&](int sl, int buf, int i) __attribute__((always_inline)) {
    const unsigned b = lds0 + (unsigned)(buf * bufF * 4);
    const int o = wv + NWAVES * (i >> 1);
    if (i & 1) {
      const bool on = sl < nslab && o < yoct;
      dma16(yr, on ? b + (xoct + o) * OCTF * 4 : lds0 + LDS_BYTES - DUMMY_BYTES, vo, on ? sl * s_rows4 * 32 + o * 1024 : 0x7ffffc00);
    } else {
      const bool on = sl < nslab && o < xoct;
      dma16(xr, on ? b + o * OCTF * 4 : lds0 + LDS_BYTES - DUMMY_BYTES, vo, on ? sl * g_rows4 * 32 + o * 1024 : 0x7ffffc00);
    }
  };
  auto issue = [&](int sl, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2 * TM / NWAVES; ++i) piece(sl, buf, i);
  };
  // wait until at most (nbuf - 2) slabs' worth of this wave's DMA instructions (16 each) are still in flight, i.e.
  // the oldest outstanding slab has landed, then publish it
  auto publish = [&]() __attribute__((always_inline)) {
    if (nbuf == 2) __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
    else if (nbuf == 3) __builtin_amdgcn_s_waitcnt(0x4f70);   // vmcnt(16)
    else __builtin_amdgcn_s_waitcnt(0x8f70);                  // vmcnt(32)
    __syncthreads();
  };

  for (int sl = 0; sl < nbuf - 1; ++sl) issue(sl, sl);   // (slabs past the end are no-ops that still count in vmcnt)
  publish();
  CN_T(0)
  if (!active) {   // idle wave of a narrow GEMM: it still moves its share of the slabs and meets the barriers
    int nb = nbuf - 1;   // buffer of slab sl + nbuf - 1
    for (int sl = 0; sl < nslab; ++sl) {
      issue(sl + nbuf - 1, nb);
      nb = nb + 1 == nbuf ? 0 : nb + 1;
      publish();
    }
    return;
  }
  int cur = 0, nb = nbuf - 1;
  for (int sl = 0; sl < nslab; ++sl) {
    // LDS image: block of column octet o at o*OCTF floats, inside it point m, column c at m*8 + c.  Lane (i, hh) of
    // step st reads column 32x + i of point 2*st + hh: one per-lane base + the immediate (4x*OCTF + 16 st) floats;
    // the 8-float pad makes the 32 lanes of a half-wave hit 32 distinct banks.
    const int lbase = (i31 >> 3) * OCTF + hh * 8 + (i31 & 7);
    const float* Xs = lds + cur * bufF + 4 * tn0 * OCTF + lbase;
    const float* Ys = lds + cur * bufF + (xoct + 4 * tk0) * OCTF + lbase;
    // operand reads run LA steps ahead of the MFMAs that consume them: one step when a step is >= 8 MFMAs (512+
    // cycles cover the LDS latency), three for the narrow GEMMs whose steps are only 1-4 MFMAs long
    constexpr int LA = AN * AK >= 8 ? 1 : 3, R = LA + 1;
    float av[R][AN], bv[R][AK];
    auto rd = [&](int st, int o) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < AN; ++x) av[o][x] = Xs[16 * st + 4 * x * OCTF];
#pragma unroll
      for (int y = 0; y < AK; ++y) bv[o][y] = Ys[16 * st + 4 * y * OCTF];
    };
#pragma unroll
    for (int st = 0; st < LA; ++st) rd(st, st);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < TM / 2; ++st) {
      const int o = st % R, nx = (st + LA) % R;
      // Per step: AN+AK LDS reads (operands of step st+LA), one DMA piece of slab sl+nbuf-1 (into the buffer slab sl-1
      // was read from: all its readers passed the previous barrier; one piece per step because a burst of 16 KiB per
      // wave exceeds what the memory pipeline accepts at once and the in-order wave then sits in the issue queue —
      // measured 25 % of the kernel), its scalar address arithmetic, and the bias adds.  A wave issues about one
      // instruction per 4 cycles, so ~25 of them between two MFMAs (64 cycles apart) leave the matrix pipe idle for
      // ~45 cycles per step (measured, 4.5 %): when a step has >= 8 MFMAs the side work is dealt out one item per MFMA.
      // The 16 pieces go out two per step in the first half of the slab: the wave waits for its own pieces before the
      // barrier that publishes the slab, and a piece issued in the last steps has not landed by then (HBM latency
      // ~2000 cycles = 2 steps; measured 900 cycles of exposed wait per slab with one piece per step).
      constexpr bool SPREAD = AN * AK >= AN + AK + 2;
      if (!SPREAD) {
        if (st + LA < TM / 2) rd(st + LA, nx);
        piece(sl + nbuf - 1, nb, st);
        __builtin_amdgcn_sched_barrier(0);
        if (BS) {
#pragma unroll
          for (int x = 0; x < AN2; ++x) bsum2[x] += f32x2{av[o][2 * x], av[o][2 * x + 1 < AN ? 2 * x + 1 : 2 * x]};
        }
      }
#pragma unroll
      for (int x = 0; x < AN; ++x)
#pragma unroll
        for (int y = 0; y < AK; ++y) {
          acc[x][y] = mfma(av[o][x], bv[o][y], acc[x][y]);
          if (SPREAD) {
            const int j = x * AK + y;
            if (st + LA < TM / 2) {
              if (j < AN) av[nx][j] = Xs[16 * (st + LA) + 4 * j * OCTF];
              else if (j < AN + AK) bv[nx][j - AN] = Ys[16 * (st + LA) + 4 * (j - AN) * OCTF];
            }
            if (BS && j < AN2) bsum2[j] += f32x2{av[o][2 * j], av[o][2 * j + 1 < AN ? 2 * j + 1 : 2 * j]};
            if (st < TM / 4 && j == AN + AK) piece(sl + nbuf - 1, nb, 2 * st);
            if (st < TM / 4 && j == AN + AK + 1) piece(sl + nbuf - 1, nb, 2 * st + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    CN_T(2)
    cur = cur + 1 == nbuf ? 0 : cur + 1;
    nb = nb + 1 == nbuf ? 0 : nb + 1;
    publish();
    CN_T(1)
  }
  // Epilogue: the wave's AN x AK accumulator tiles -> this split's slice of the partial buffer, 256 dword stores per
  // lane.  Buffer stores against a resource that covers exactly the job's (N - n_lo) x ld rows: addresses are one
  // per-lane VGPR (column, half-wave row) + a scalar row offset; lanes past K carry an out-of-range offset and are
  // dropped by the hardware; nothing waits on anything.  (__float_as_uint, not __builtin_bit_cast: the latter applied to a
  // vector ELEMENT reads element 0.)  (As plain pointer stores these were `flat_store` + a scratch reload of the
  // job bounds + s_waitcnt vmcnt(0) EACH — the argument struct lives in scratch, a flat store may alias it — i.e. 256
  // serialised round trips, 2.8 % of a 256x256 job.)
  const int N = __builtin_amdgcn_readfirstlane(jb.N), K = __builtin_amdgcn_readfirstlane(jb.K);
  const int n_lo = __builtin_amdgcn_readfirstlane(jb.n_lo), ld4 = __builtin_amdgcn_readfirstlane(jb.ld * 4);
  const int col0 = __builtin_amdgcn_readfirstlane(jb.col0);
  // (values read from the scratch-resident argument struct count as divergent: make the pointers provably uniform,
  // or every store becomes a waterfall loop over the "different" resources)
  auto uniform = [](const float* q) __attribute__((always_inline)) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
  };
  const float* out = a.partials + (int64_t)split * a.pstride;
  const rsrc_t wr = make_rsrc(uniform(out + a.toff[jb.tensor]), (unsigned)((N - n_lo) * ld4));
  int kvo[AK];   // byte offset of (row 4*hh, column k) or an always-out-of-range one for k >= K
#pragma unroll
  for (int y = 0; y < AK; ++y) {
    const int k = 32 * (tk0 + y) + i31;
    kvo[y] = k < K ? (col0 + k) * 4 + hh * 4 * ld4 : TM_OOB;
  }
#pragma unroll
  for (int x = 0; x < AN; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n0 = 32 * (tn0 + x) + (r & 3) + 8 * (r >> 2);   // row of the hh = 0 half-wave (scalar); hh = 1: n0 + 4
      if (n0 >= N) continue;
      if (n0 >= n_lo && n0 + 4 < N) {   // both half-waves' rows valid (wave-uniform test)
#pragma unroll
        for (int y = 0; y < AK; ++y)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[x][y][r]), wr, kvo[y], (n0 - n_lo) * ld4, 0);
      } else {   // ragged heads (N = 3, 4, 5; the sigma head starts at n_lo = 3): per-lane row test
        const int n = n0 + 4 * hh;
#pragma unroll
        for (int y = 0; y < AK; ++y) {
          const int k = 32 * (tk0 + y) + i31;
          const int vo = (n >= n_lo && n < N && k < K) ? ((n - n_lo) * (ld4 >> 2) + col0 + k) * 4 : TM_OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[x][y][r]), wr, vo, 0, 0);
        }
      }
    }
  CN_T(3)
  CN_TEND
  if (BS) {
    const rsrc_t br = make_rsrc(uniform(out + a.toff[jb.bias_tensor]), (unsigned)((N - n_lo) * 4));
#pragma unroll
    for (int x = 0; x < AN; ++x) {
      const float bx = bsum2[x >> 1][x & 1];
      const float s = bx + __shfl_xor(bx, 32, 64);
      const int n = 32 * (tn0 + x) + i31;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), br, (hh == 0 && n >= n_lo && n < N) ? (n - n_lo) * 4 : TM_OOB, 0, 0);
    }
  }
}

// ---- OPT-IN bf16x3 body (second bench line only) -----------------------------------------------------------------------
// The same GEMM, point ranges, LDS image, DMA and epilogue, with the products on v_mfma_f32_32x32x16_bf16: a K-step contracts 16
// points (lane (i, hh) supplies column i of points 8 hh + e, e = 0..7: eight ds_read_b32 at immediate offsets), every fragment —
// 32 columns x 16 points of dZ or of H, fp32 in LDS — is split into THREE bf16 planes on the VALU (mlp_bf_common.hpp split_pair)
// and each 32x32 tile pair takes the 6 cross terms with i + j < 3 (error per product ~2^-23: fp32-equivalent, fp32 accumulate).
// Pipeline: the planes of K-step kk+1 are produced (reads + splits, dealt out two half-fragments per tile pair) under the 6 AN AK
// MFMAs of K-step kk; a slab (32 points = 2 K-steps) is published by ONE barrier in the MIDDLE of the previous slab's iteration,
// right before its first reads; its DMA is issued a full iteration ahead (two buffers).
template <int AN, int AK, bool BS>
__device__ __forceinline__ void wgrad_body_bf3(WgNetC& a, WgJobC& jb, float* lds, const int vb) {
  constexpr int NF = AN + AK;                       // fragments per K-step: A (dZ columns) 0..AN-1, B (H columns) AN..NF-1
  static_assert(AN * AK >= NF, "side work is dealt out over the tile pairs");
  const int split = __builtin_amdgcn_readfirstlane(vb - jb.first);
  const int tid = threadIdx.x, lane = tid & 63, i31 = lane & 31, hh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv / jb.gk, wk = wv - wn * jb.gk;
  const int g_rows4 = __builtin_amdgcn_readfirstlane(a.g_rows * 4), s_rows4 = __builtin_amdgcn_readfirstlane(a.s_rows * 4);
  const int64_t m_begin0 = (int64_t)split * jb.chunk;
  const int64_t m_begin = m_begin0 < a.Mp ? m_begin0 : a.Mp;
  const int64_t m_end = m_begin + jb.chunk < a.Mp ? m_begin + jb.chunk : a.Mp;
  const int nslab = __builtin_amdgcn_readfirstlane(m_end > m_begin ? (int)((m_end - m_begin) / TM) : 0);
  const int ntn = (jb.N + 31) >> 5, ntk = (jb.K + 31) >> 5;
  const int tn0 = __builtin_amdgcn_readfirstlane(wn * AN), tk0 = __builtin_amdgcn_readfirstlane(wk * AK);
  const i32x4 xr = dma_rsrc(a.G + m_begin * a.g_rows + (jb.xcol >> 3) * 256, (unsigned)((m_end - m_begin) * a.g_rows * 4));
  const i32x4 yr = dma_rsrc(a.stash + m_begin * a.s_rows + (jb.ycol >> 3) * 256, (unsigned)((m_end - m_begin) * a.s_rows * 4));
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds);
  const int vo = lane * 16;
  f32x16 acc[AN][AK];
#pragma unroll
  for (int x = 0; x < AN; ++x)
#pragma unroll
    for (int y = 0; y < AK; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float bsum[AN];
#pragma unroll
  for (int x = 0; x < AN; ++x) bsum[x] = 0.f;
  const int xoct = __builtin_amdgcn_readfirstlane(4 * ntn), yoct = __builtin_amdgcn_readfirstlane(4 * ntk);
  const int bufF = (xoct + yoct) * OCTF;            // two buffers (the envelope check guarantees they fit)
  auto piece = [&](int sl, int buf, int i) __attribute__((always_inline)) {
    const unsigned b = lds0 + (unsigned)(buf * bufF * 4);
    const int o = wv + NWAVES * (i >> 1);
    if (i & 1) {
      const bool on = sl < nslab && o < yoct;
      dma16(yr, on ? b + (xoct + o) * OCTF * 4 : lds0 + LDS_BYTES - DUMMY_BYTES, vo, on ? sl * s_rows4 * 32 + o * 1024 : 0x7ffffc00);
    } else {
      const bool on = sl < nslab && o < xoct;
      dma16(xr, on ? b + o * OCTF * 4 : lds0 + LDS_BYTES - DUMMY_BYTES, vo, on ? sl * g_rows4 * 32 + o * 1024 : 0x7ffffc00);
    }
  };
  auto landed = [&]() __attribute__((always_inline)) {   // this wave's DMA pieces have landed; everybody's: the barrier
#ifndef CN_ABL_NOWAIT                                    // (ablation builds, timings only: -DCN_ABL_NOWAIT / NOSPLIT / NOREAD / NOBAR)
    __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0)
#endif
#ifndef CN_ABL_NOBAR
    __syncthreads();
#endif
  };
  // LDS image (as wgrad_body): block of column octet o at o*OCTF floats, inside it point p, column c at p*8 + c.  Lane (i, hh)
  // reads column 32x + i of points 16 ks + 8 hh + e: one per-lane base + the immediate (4x*OCTF + 128 ks + 8 e) floats; the 32
  // lanes of a half-wave hit 32 distinct banks (the half-waves are served in separate cycles).
  const int lbase = (i31 >> 3) * OCTF + (i31 & 7) + hh * 64;
  u32x4 pl[2][NF][3];     // bf16 planes of the fragments: set kk & 1 feeds the MFMAs of K-step kk
  constexpr int FPR = (2 * NF + AN * AK - 1) / (AN * AK);   // fragments split per PAIR of tile pairs (= half-fragment items per slot)
  float rw[2 * FPR][8];   // raw fp32 values of the fragments in flight: those being split and those being read (one pair ahead)
  // fragment f of K-step ks of the slab in buffer `buf`: its eight values -> rw[f & 1]
  auto rd = [&](int buf, int ks, int f, int half) __attribute__((always_inline)) {
    const float* base = lds + buf * bufF + (f < AN ? 4 * (tn0 + f) : xoct + 4 * (tk0 + f - AN)) * OCTF + lbase + 128 * ks;
#pragma unroll
    for (int e = 4 * half; e < 4 * half + 4; ++e) rw[f & 1][e] = base[8 * e];
  };
  auto sp = [&](int set, int f, int half) __attribute__((always_inline)) {
    float* v = rw[f & 1];
    if (BS && f < AN) bsum[f] += (v[4 * half] + v[4 * half + 1]) + (v[4 * half + 2] + v[4 * half + 3]);
    split_pair<3, false>(v[4 * half], v[4 * half + 1], pl[set][f], 2 * half);
    split_pair<3, false>(v[4 * half + 2], v[4 * half + 3], pl[set][f], 2 * half + 1);
  };
  // produce all planes of one K-step with nothing to hide behind (prologue only)
  auto produce_now = [&](int set, int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      rd(buf, ks, f, 0); rd(buf, ks, f, 1);
      sp(set, f, 0); sp(set, f, 1);
    }
  };
  // MFMAs of the K-step whose planes are in `set`, with the production of the next K-step's planes (from buffer nbuf_, K-step
  // nks of that slab, into set ^ 1) dealt out as half-fragment items: item h reads half (h & 1) of fragment h/2 + 1 and splits
  // half (h & 1) of fragment h/2 (read one fragment earlier); fragment 0 is read in front of the loop.  HPS items ride behind
  // every tile pair, and inside a tile pair every instruction of an item sits in the gap behind ONE of the six MFMAs (a wave
  // issues in order and the six cross terms are a dependent chain: work placed between them is free, a burst behind them is not):
  //   gap 0: reads e0, e1 of the next fragment; stage 0 of both register pairs       gap 1: reads e2, e3; residual of pair A
  //   gap 2: plane 1 of A, residual of B       gap 3: plane 1 of B, residual of A, the DMA pieces       gap 4: plane 2 of A,
  //   residual of B       gap 5: plane 2 of B, the bias column sums.
  // DMA: pieces of slab dsl -> buffer dbuf, two per tile pair in the first eight (dsl < 0: none).
  constexpr int IA[6] = {0, 1, 2, 0, 1, 0}, IB[6] = {2, 1, 0, 1, 0, 0};
  // (prod / dma are COMPILE-TIME flags — std::integral_constant arguments — so that the two phases of the slab loop are
  //  straight-line code: with run-time flags the compiler cut the loop into dozens of basic blocks of a few MFMAs each)
  auto phase = [&](int set, auto prod_c, int nbuf_, int nks, auto dma_c, int dsl, int dbuf) __attribute__((always_inline)) {
    constexpr bool prod = decltype(prod_c)::value, dma = decltype(dma_c)::value;
    constexpr int HPS = FPR;
    if (prod) {
#pragma unroll
      for (int f = 0; f < FPR; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          rw[f][e] = (lds + nbuf_ * bufF + (f < AN ? 4 * (tn0 + f) : xoct + 4 * (tk0 + f - AN)) * OCTF + lbase + 128 * nks)[8 * e];
    }
    __builtin_amdgcn_sched_barrier(0);
    static_assert(AK % 2 == 0, "tile pairs");
#pragma unroll
    for (int x = 0; x < AN; ++x)
#pragma unroll
      for (int y0 = 0; y0 < AK; y0 += 2) {
        // two tile pairs at a time, their six-MFMA chains alternating: consecutive MFMAs never share an accumulator
        Split3 SA2[2][HPS], SB2[2][HPS];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          const int y = y0 + uu, j = x * AK + y;
          Split3 (&SA)[HPS] = SA2[uu];
          Split3 (&SB)[HPS] = SB2[uu];
          acc[x][y] = mfma_bf(pl[set][x][IA[k]], pl[set][AN + y][IB[k]], acc[x][y]);
          if (prod) {
#pragma unroll
            for (int u = 0; u < HPS; ++u) {
              const int h = j * HPS + u;
              if (h >= 2 * NF) continue;
              const int f = h >> 1, hf = h & 1, fn = f + FPR;     // fn: the fragment read now, split one pair of tile pairs later
              float* v = rw[f % (2 * FPR)];
#ifdef CN_ABL_NOREAD
              if (false) {
#else
              if (fn < NF && k < 2) {        // its values e = 4 hf + 2 k, + 1
#endif
                const float* base = lds + nbuf_ * bufF + (fn < AN ? 4 * (tn0 + fn) : xoct + 4 * (tk0 + fn - AN)) * OCTF + lbase + 128 * nks;
                rw[fn % (2 * FPR)][4 * hf + 2 * k] = base[8 * (4 * hf + 2 * k)];
                rw[fn % (2 * FPR)][4 * hf + 2 * k + 1] = base[8 * (4 * hf + 2 * k + 1)];
              }
#ifdef CN_ABL_NOSPLIT
              if (k == 0 && false) {
#else
              if (k == 0) {
#endif
                split3_s0<false>(SA[u], v[4 * hf], v[4 * hf + 1], pl[set ^ 1][f], 2 * hf);
                split3_s0<false>(SB[u], v[4 * hf + 2], v[4 * hf + 3], pl[set ^ 1][f], 2 * hf + 1);
              }
#ifndef CN_ABL_NOSPLIT
              if (k == 1 || k == 3) split3_residual(SA[u]);
              if (k == 2 || k == 4) split3_residual(SB[u]);
              if (k == 2) split3_plane(SA[u], pl[set ^ 1][f], 1, 2 * hf);
              if (k == 3) split3_plane(SB[u], pl[set ^ 1][f], 1, 2 * hf + 1);
              if (k == 4) split3_plane(SA[u], pl[set ^ 1][f], 2, 2 * hf);
#endif
              if (k == 5) {
#ifndef CN_ABL_NOSPLIT
                split3_plane(SB[u], pl[set ^ 1][f], 2, 2 * hf + 1);
#endif
                if (BS && f < AN) bsum[f] += (v[4 * hf] + v[4 * hf + 1]) + (v[4 * hf + 2] + v[4 * hf + 3]);
              }
            }
          }
          // the 16 DMA pieces of the next slab: two per tile pair in the first eight (>= 8 tile pairs per K-step), four per tile
          // pair when the wave owns only four (the 2 x 2 narrow GEMMs)
          if (AN * AK >= 8) {
            if (dma && j < 8 && k == 3) { piece(dsl, dbuf, 2 * j); piece(dsl, dbuf, 2 * j + 1); }
          } else {
            if (dma && k == 1) { piece(dsl, dbuf, 4 * j); piece(dsl, dbuf, 4 * j + 1); }
            if (dma && k == 3) { piece(dsl, dbuf, 4 * j + 2); piece(dsl, dbuf, 4 * j + 3); }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  // prologue: slab 0 lands, slab 1 is on its way, the planes of K-step 0 are produced in the open
#pragma unroll
  for (int i = 0; i < 2 * TM / NWAVES; ++i) piece(0, 0, i);
  landed();
#pragma unroll
  for (int i = 0; i < 2 * TM / NWAVES; ++i) piece(1, 1, i);
  if (nslab > 0) produce_now(0, 0, 0);
  int cur = 0;
  const std::true_type yes{};
  const std::false_type no{};
  for (int sl = 0; sl + 1 < nslab; ++sl) {
    phase(0, yes, cur, 1, no, 0, 0);                     // K-step 2 sl; planes of 2 sl + 1 from the same slab
    landed();                                            // slab sl + 1 has landed, nobody reads slab sl any more
    phase(1, yes, cur ^ 1, 0, yes, sl + 2, cur);         // K-step 2 sl + 1; planes of 2 sl + 2 from slab sl + 1; DMA of slab sl + 2
    cur ^= 1;
  }
  if (nslab > 0) {                                       // the last slab: nothing follows
    phase(0, yes, cur, 1, no, 0, 0);
    landed();
    phase(1, no, cur, 0, no, 0, 0);
  }
  // Epilogue: as wgrad_body (the accumulator layout of a 32x32 MFMA tile is the same)
  const int N = __builtin_amdgcn_readfirstlane(jb.N), K = __builtin_amdgcn_readfirstlane(jb.K);
  const int n_lo = __builtin_amdgcn_readfirstlane(jb.n_lo), ld4 = __builtin_amdgcn_readfirstlane(jb.ld * 4);
  const int col0 = __builtin_amdgcn_readfirstlane(jb.col0);
  auto uniform = [](const float* q) __attribute__((always_inline)) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
  };
  const float* out = a.partials + (int64_t)split * a.pstride;
  const rsrc_t wr = make_rsrc(uniform(out + a.toff[jb.tensor]), (unsigned)((N - n_lo) * ld4));
  int kvo[AK];
#pragma unroll
  for (int y = 0; y < AK; ++y) {
    const int k = 32 * (tk0 + y) + i31;
    kvo[y] = k < K ? (col0 + k) * 4 + hh * 4 * ld4 : TM_OOB;
  }
#pragma unroll
  for (int x = 0; x < AN; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n0 = 32 * (tn0 + x) + (r & 3) + 8 * (r >> 2);
      if (n0 >= N) continue;
      if (n0 >= n_lo && n0 + 4 < N) {
#pragma unroll
        for (int y = 0; y < AK; ++y)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[x][y][r]), wr, kvo[y], (n0 - n_lo) * ld4, 0);
      } else {
        const int n = n0 + 4 * hh;
#pragma unroll
        for (int y = 0; y < AK; ++y) {
          const int k = 32 * (tk0 + y) + i31;
          const int vo2 = (n >= n_lo && n < N && k < K) ? ((n - n_lo) * (ld4 >> 2) + col0 + k) * 4 : TM_OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[x][y][r]), wr, vo2, 0, 0);
        }
      }
    }
  if (BS) {
    const rsrc_t br = make_rsrc(uniform(out + a.toff[jb.bias_tensor]), (unsigned)((N - n_lo) * 4));
#pragma unroll
    for (int x = 0; x < AN; ++x) {
      const float s = bsum[x] + __shfl_xor(bsum[x], 32, 64);
      const int n = 32 * (tn0 + x) + i31;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), br, (hh == 0 && n >= n_lo && n < N) ? (n - n_lo) * 4 : TM_OOB, 0, 0);
    }
  }
}

template <int AN, int AK>
__device__ __forceinline__ void wgrad_disp_bf3(WgNetC& a, WgJobC& jb, float* lds, const int vb) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (jb.bias_tensor >= 0 && wv % jb.gk == 0) wgrad_body_bf3<AN, AK, true>(a, jb, lds, vb);
  else wgrad_body_bf3<AN, AK, false>(a, jb, lds, vb);
}

template <int AN, int AK>
__device__ __forceinline__ void wgrad_disp(WgNetC& a, WgJobC& jb, float* lds, const int vb) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (jb.bias_tensor >= 0 && wv % jb.gk == 0) wgrad_body<AN, AK, true>(a, jb, lds, vb);
  else wgrad_body<AN, AK, false>(a, jb, lds, vb);
}

// `wgrad_k` = the exact-fp32 kernel of the default path (its code is untouched by the opt-in arithmetic); `wgrad_mixed_k` = the
// same grid with the bf16x3 body for the jobs flagged bf3 and the fp32 body for the rest (launched only by cnerf_mlp_wgrad_bf*).
template <bool MIXED>
__device__ __forceinline__ void wgrad_one(WgArgsC& args, float* lds, const int vb) {
  // 1-D grid, jobs back to back: the job of this block = the last one whose first block id is <= vb (scalar loads
  // from the kernarg segment, <= 28 entries)
  int ji = 0;
  for (int i = 1; i < args.nj; ++i) ji = vb >= args.job[i].first ? i : ji;
  WgJobC& jb = args.job[ji];
  WgNetC& a = args.net[jb.net];
  if (MIXED && jb.bf3) {           // block-uniform: the opt-in bf16x3 body (wide GEMMs only, plan in add_net_jobs)
    switch (jb.an * 8 + jb.ak) {
      case 4 * 8 + 4: wgrad_disp_bf3<4, 4>(a, jb, lds, vb); return;
      case 4 * 8 + 2: wgrad_disp_bf3<4, 2>(a, jb, lds, vb); return;
      case 2 * 8 + 4: wgrad_disp_bf3<2, 4>(a, jb, lds, vb); return;
      case 2 * 8 + 2: wgrad_disp_bf3<2, 2>(a, jb, lds, vb); return;
      default: break;
    }
  }
  switch (jb.an * 8 + jb.ak) {     // block-uniform
    case 4 * 8 + 4: wgrad_disp<4, 4>(a, jb, lds, vb); break;
    case 4 * 8 + 2: wgrad_disp<4, 2>(a, jb, lds, vb); break;
    case 2 * 8 + 4: wgrad_disp<2, 4>(a, jb, lds, vb); break;
    case 4 * 8 + 1: wgrad_disp<4, 1>(a, jb, lds, vb); break;
    case 1 * 8 + 4: wgrad_disp<1, 4>(a, jb, lds, vb); break;
    case 2 * 8 + 2: wgrad_disp<2, 2>(a, jb, lds, vb); break;
    case 2 * 8 + 1: wgrad_disp<2, 1>(a, jb, lds, vb); break;
    case 1 * 8 + 2: wgrad_disp<1, 2>(a, jb, lds, vb); break;
    default: wgrad_disp<1, 1>(a, jb, lds, vb); break;
  }
}

template <bool MIXED>
__device__ __forceinline__ void wgrad_kernel_body() {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 buffers][X slab | Y slab]
  WgArgsC& args = *(WgArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef CN_WGRAD_DYN
  // EXPERIMENT (-DCN_WGRAD_DYN): one resident workgroup per CU pulls virtual block ids from a device counter (first id = blockIdx.x,
  // then gridDim.x + ticket): the hardware's first-free-CU balance without the workgroup hand-over (LDS release, dispatch, kernarg
  // loads).  The next ticket is requested at the START of a job (the atomic's round trip hides behind the job) and published to
  // the other waves through the LDS word in front of the DMA dummy block at its end.
  volatile int* tick = (volatile int*)(lds + LDS_FLOATS);
  int vb = (int)blockIdx.x;
  for (;;) {
    int nxt = 0;
    if (threadIdx.x == 0) nxt = (int)gridDim.x + atomicAdd(args.counter, 1);
    wgrad_one<MIXED>(args, lds, vb);
    if (threadIdx.x == 0) *tick = nxt;
    __syncthreads();
    vb = __builtin_amdgcn_readfirstlane(*tick);
    __syncthreads();
    if (vb >= args.total) break;
  }
#else
  wgrad_one<MIXED>(args, lds, (int)blockIdx.x);
#endif
}

__global__ __launch_bounds__(64 * NWAVES) void wgrad_k(WgArgs a_by_value) {
  (void)a_by_value;   // (the first and only explicit kernel argument: offset 0 of the kernarg segment, read in place)
  wgrad_kernel_body<false>();
}
__global__ __launch_bounds__(64 * NWAVES) void wgrad_mixed_k(WgArgs a_by_value) {
  (void)a_by_value;
  wgrad_kernel_body<true>();
}

struct RedArgs {    // entries [0, nt0) belong to the first network, [nt0, nt0 + nt1) to the second
  float* grad[2 * CNERF_MAX_TENSORS];
  const float* part[2 * CNERF_MAX_TENSORS];   // the tensor's slot in split slice 0
  int64_t numel[2 * CNERF_MAX_TENSORS];
  int64_t pstride[2 * CNERF_MAX_TENSORS];
  int nsplit[2 * CNERF_MAX_TENSORS];
  int touched[2 * CNERF_MAX_TENSORS];
  int accumulate;
};

// Fixed-order sum of the split partials (bit-reproducible).  Bandwidth-bound (nsplit x ~2.4 MB): 16-byte loads, 8
// independent partial streams in flight per thread, enough blocks to cover the chip.
__global__ __launch_bounds__(256) void wgrad_reduce_k(RedArgs a_by_value) {
  (void)a_by_value;   // read in place (kernarg segment, scalar loads): indexed by blockIdx.y a by-value copy lives in scratch
  const CN_CONST RedArgs& a = *(const CN_CONST RedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int t = blockIdx.y;
  float* g = a.grad[t];
  if (g == nullptr) return;
  const int64_t n = a.numel[t], n4 = n >> 2;
  const float* p0 = a.part[t];   // tensor offsets are multiples of 4 floats, pstride of 64
  const int nsplit = a.nsplit[t];
  const int64_t pstride = a.pstride[t];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (a.touched[t]) {
      const f32x4* p = reinterpret_cast<const f32x4*>(p0) + i;
      const int64_t st = pstride >> 2;
      int k = 0;
      for (; k + 8 <= nsplit; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];   // in split order
      }
      for (; k < nsplit; ++k) s += p[(int64_t)k * st];
    }
    f32x4* gp = reinterpret_cast<f32x4*>(g) + i;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
      *gp = a.accumulate ? *gp + s : s;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) g[4 * i + c] = a.accumulate ? g[4 * i + c] + s[c] : s[c];
    }
  }
  // tail (numel not a multiple of 4)
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    if (a.touched[t])
      for (int k = 0; k < nsplit; ++k) s += p0[i + (int64_t)k * pstride];
    g[i] = a.accumulate ? g[i] + s : s;
  }
}

}  // namespace

#ifdef CN_TIMING
CN_TIMING_ACCESSOR(cnerf_debug_timing_wgrad)
#endif

int64_t cn_param_floats(const NetGeom& g) {
  cnerf_net net{g.D, g.W, g.L, g.Ld, g.viewdirs, g.out_ch, g.skip};
  // note: g.skip was normalised to -1 when inactive; tensor shapes only depend on the active skip
  int64_t tot = 0;
  const int nt = cnerf_num_tensors(&net);
  for (int i = 0; i < nt; ++i) {
    int64_t r, c;
    cnerf_tensor_shape(&net, i, &r, &c);
    tot += cn_round_up(r * c, 4);
  }
  return tot;
}

int cn_wgrad_nsplit(int64_t Mp) {
  // CAPACITY of the partial-gradient buffer in slices (each cn_param_floats(g) ~ 2.4 MB at D=8/W=256): the most point ranges
  // any GEMM of this network may be cut into — ranges of >= 16 slabs (512 points), at most 128 of them, and <= 65536 points
  // per range (a range's rows stay within 32-bit byte offsets).  How many each GEMM actually gets is planned per launch
  // (plan_ranges); the buffer is sized for the cap.
  int64_t s = Mp / 512;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  if (s * 65536 < Mp) s = (Mp + 65535) / 65536;
  return (int)s;
}

namespace {

// Appends one network's GEMMs to the job table / reduction table.  Returns false when a shape is outside the envelope.
bool add_net_jobs(const NetGeom& g, int netidx, const float* stash, const float* G, int64_t Mp, float* partials,
                  const cnerf_ptrs* grads, WgArgs& a, int& nj, RedArgs& r, int& nr, bool bf3 = false) {
  cnerf_net net{g.D, g.W, g.L, g.Ld, g.viewdirs, g.out_ch, g.skip};
  WgNet& wn = a.net[netidx];
  const int nt = cnerf_num_tensors(&net);
  const int r0 = nr;
  int64_t off = 0;
  for (int i = 0; i < nt; ++i) {
    int64_t rr, cc;
    cnerf_tensor_shape(&net, i, &rr, &cc);
    wn.toff[i] = off;
    r.numel[r0 + i] = rr * cc;
    r.grad[r0 + i] = grads->p[i];
    r.part[r0 + i] = partials + off;
    r.nsplit[r0 + i] = 0;
    r.touched[r0 + i] = 0;
    off += cn_round_up(rr * cc, 4);
  }
  nr += nt;
  const int64_t pstride = cn_round_up(off, 64);
  for (int i = 0; i < nt; ++i) r.pstride[r0 + i] = pstride;
  const int D = g.D, W = g.W, Wh = g.Wh;
  bool ok = true;
  auto add = [&](int xcol, int ycol, int N, int K, int tensor, int ld, int col0, int bias_tensor, int n_lo = 0) {
    const int ntn = (N + 31) / 32, ntk = (K + 31) / 32;
    // wave grid gn x gk in {1x4, 2x2, 4x1}: an x ak <= 4x4 tiles per wave; minimise the busiest wave's tile
    // count (= the workgroup's MFMA time), then the operand traffic an+ak
    int best_gk = 0, best_an = 0, best_ak = 0, best_cost = 1 << 30;
    for (int gn = 1; gn <= NWAVES; gn *= 2) {
      const int gk = NWAVES / gn;
      const int an = (ntn + gn - 1) / gn, ak = (ntk + gk - 1) / gk;
      if (an > 4 || ak > 4 || an * ak > 64 / NWAVES) continue;   // <= 256 accumulator registers per wave (128 at 8 waves)
      const int cost = an * ak * 16 + an + ak;
      if (cost < best_cost) { best_cost = cost; best_gk = gk; best_an = an; best_ak = ak; }
    }
    if (best_gk == 0 || best_an == 3 || best_ak == 3 || ntn > 8 || ntk > 8 || nj >= MAX_WG_JOBS) { ok = false; return; }
    // the opt-in bf16x3 body takes the wide GEMMs (>= 8 tiles per wave: 86 % of the MACs at D=8/W=256) whose two slab buffers
    // fit the LDS; the narrow ones (heads, gamma columns) stay exact fp32 in the same grid
    // (round 5: the 2 x 2 jobs — the two 256 x 63 GEMMs of the encoding columns, 62 % of the narrow GEMMs' CU time — CAN take the
    //  bf16x3 body (CNERF_BF3_NARROW=1).  Measured level: wgrad 5.97-6.01 ms with, 5.98-6.06 without (profiles/r05_bf3_narrow_ab.txt):
    //  with two slab buffers these short jobs wait for their DMA whatever the arithmetic.  Default off: they stay exact fp32.)
    static const bool narrow22 = getenv("CNERF_BF3_NARROW") && atoi(getenv("CNERF_BF3_NARROW")) != 0;
    const bool wide = (best_an == 4 && best_ak == 4) || (best_an == 4 && best_ak == 2) || (best_an == 2 && best_ak == 4) ||
                      (narrow22 && best_an == 2 && best_ak == 2);
    const bool fits = 2 * (4 * ntn + 4 * ntk) * OCTF <= LDS_FLOATS;
    a.job[nj++] = WgJob{netidx, xcol, ycol, N, K, n_lo, tensor, ld, col0, bias_tensor, best_gk ? best_gk : 2, best_an,
                        best_ak, 1, 0, 0, (bf3 && wide && fits) ? 1 : 0};
    r.touched[r0 + tensor] = 1;
    if (bias_tensor >= 0) r.touched[r0 + bias_tensor] = 1;
  };
  const int base = 2 * D;
  for (int l = 1; l < D; ++l) {
    const bool sk = g.skip >= 0 && l == g.skip + 1;
    add(g.g_z[l], g.s_h[l - 1], W, W, 2 * l, sk ? W + g.in_ch : W, sk ? g.in_ch : 0, 2 * l + 1);
  }
  if (g.viewdirs) {
    add(g.g_feat, g.s_h[D - 1], W, W, base + 2, W, 0, base + 3);
    add(g.g_hv, g.s_feat, Wh, W, base + 0, W + g.dir_ch, 0, base + 1);
  }
  add(g.g_z[0], g.s_enc, W, g.in_ch, 0, g.in_ch, 0, 1);
  if (g.skip >= 0) add(g.g_z[g.skip + 1], g.s_enc, W, g.in_ch, 2 * (g.skip + 1), W + g.in_ch, 0, -1);
  if (g.viewdirs) {
    add(g.g_hv, g.s_denc, Wh, g.dir_ch, base + 0, W + g.dir_ch, W, -1);
    add(g.g_out, g.s_h[D - 1], 4, W, base + 4, W, 0, base + 5, 3);   // d_sigma is column 3 of the d_raw copy
    add(g.g_out, g.s_hv, 3, Wh, base + 6, Wh, 0, base + 7);
  } else {
    add(g.g_out, g.s_h[D - 1], g.out_ch, W, base + 2, W, 0, base + 3);
  }
  wn.stash = stash; wn.G = G; wn.partials = partials; wn.Mp = Mp; wn.pstride = pstride;
  wn.s_rows = g.s_rows; wn.g_rows = g.g_rows;
  wn.live = nullptr; wn.live_mul = 0; wn.live_sub = 0;
  return ok;
}

// CNERF_WGRAD_NSPLIT="a" or "a,b": tuning knob (scripts/kbench_pair.py) — every GEMM of network 0 (and 1) gets exactly a (b) ranges
void forced_counts(int* f) {
  f[0] = f[1] = 0;
  if (const char* e = getenv("CNERF_WGRAD_NSPLIT")) {
    f[0] = f[1] = atoi(e);
    if (const char* c = strchr(e, ',')) f[1] = atoi(c + 1);
  }
}

// ---- how many point ranges each GEMM gets -------------------------------------------------------------------------------
// A workgroup occupies a whole CU (160 KiB of LDS) for (its range's slabs) x (its GEMM's time per slab: 1024 cycles per 32x32
// tile of the busiest wave + ~560 of barrier — measured 17000-17220 / 8830-8920 / 4705-4860 / 2563-2582 / 1540-1680 cycles
// for 16 / 8 / 4 / 2 / 1 tiles, scripts/wgrad_trace.py); blocks go to the 8 XCDs round-robin in grid order (block i runs on
// XCC i % 8, measured), each XCD hands its blocks to its first free CU; the launch is over when the last one finishes.
// Rounds 1-2 cut every GEMM of a network into Mp / 4096 ranges: right at 4096 rays (128 + 64 ranges, 10 rounds), ruinous at the
// 512 rays per GPU of the 8-way strong-scaling shard (24 + 8 ranges: the 256 long workgroups are exactly ONE round at 0.95 ms,
// the 192 short ones a second, mostly empty one — 1.88 ms for 1.0 ms of MFMA work at peak).  A dispatch simulator fed with the
// measured slab costs predicts a plan's makespan to ~2 % at a fixed clock, but could not rank plans that close: the clock the
// chip sustains depends on the mix in flight (equal-duration plans that the simulator preferred ran 5-15 % slower than plans of
// many unequal workgroups).  So the rule is read off a measured grid sweep at 512 / 1024 / 4096 rays (fine x coarse counts,
// profiles/r03_wgrad_grid.txt; flat within ~1 % for counts of 32-64, cliffs when the smaller network gets < 1/4 of the larger
// one's count): 64 ranges for a network of >= 3072 slabs (98 304 points), 32 for a smaller one, never shorter than 16 slabs,
// never beyond the partial buffer's capacity — (64, 32) at 512 and 1024 rays, (64, 64) from 1536 rays up.  The count depends on
// the network's OWN point count only, so the merged coarse+fine launch sums its partials in exactly the order of two separate
// launches (bit-identical gradients, tests), and every GEMM of a network shares it (GEMMs that write columns of one parameter
// tensor must: one reduction per tensor).
void plan_ranges(WgArgs& a, int nj, const int* cap, int* ns_out) {
  int forced[2];
  forced_counts(forced);
  for (int i = 0; i < nj; ++i) {
    const int net = a.job[i].net;
    const int64_t slabs = a.net[net].Mp / TM;
    int v = slabs >= 3072 ? 64 : (slabs >= 512 ? 32 : (int)(slabs / 16));
    if (forced[0] > 0) v = forced[net] > 0 ? forced[net] : forced[0];
    const int64_t vmin = (slabs * TM + 98303) / 98304;   // <= 98304 points per range: 32-bit byte offsets of a range's rows
    if (v < vmin) v = (int)vmin;
    if (v > cap[net]) v = cap[net];
    if ((int64_t)v > slabs) v = (int)slabs;
    if (v < 1) v = 1;
    // no empty trailing ranges: with ranges of ceil(slabs / v) slabs, only ceil(slabs / that) of them hold points
    const int64_t per = (slabs + v - 1) / v;
    if (per > 0) v = (int)((slabs + per - 1) / per);
    ns_out[i] = v < 1 ? 1 : v;
  }
}

void order_jobs(const WgJob* job, int nj, const int* ns, const int64_t* slabs, int* order) {
  // longest workgroups first (stable): the short ones fill the tail
  for (int i = 0; i < nj; ++i) order[i] = i;
  auto t = [&](int i) { return (double)((slabs[i] + ns[i] - 1) / ns[i]) * (1024.0 * job[i].an * job[i].ak + 560.0); };
  for (int i = 1; i < nj; ++i) {
    const int v = order[i];
    int k = i - 1;
    for (; k >= 0 && t(order[k]) < t(v); --k) order[k + 1] = order[k];
    order[k + 1] = v;
  }
}

}  // namespace

// Weight gradients of one network (n = 1) or of two independent ones in one grid (n = 2; cnerf_mlp_bwd_pair).  `nsplit` is the
// capacity of each network's partial buffer in slices (cn_wgrad_nsplit).
int cn_wgrad_launch_n(int n, const NetGeom* const* g, const float* const* stash, const float* const* G, const int64_t* Mp,
                      float* const* partials, const int* nsplit, const cnerf_ptrs* const* grads, int accumulate,
                      hipStream_t st, int bf3, const int* live, const int* live_mul, const int* live_sub) {
  WgArgs a;
  RedArgs r;
  int nj = 0, nr = 0, r0[2] = {0, 0};
  if (live && (bf3 || !live_mul)) return CNERF_E_UNSUPPORTED;     // (the device-side row count is the exact-fp32 body's)
  for (int i = 0; i < n; ++i) {
    r0[i] = nr;
    if (!add_net_jobs(*g[i], i, stash[i], G[i], Mp[i], partials[i], grads[i], a, nj, r, nr, bf3 != 0)) return CNERF_E_UNSUPPORTED;
    a.net[i].live = live;
    a.net[i].live_mul = live ? live_mul[i] : 0;
    a.net[i].live_sub = (live && live_sub) ? live_sub[i] : 0;
    if (live && (live_mul[i] <= 0 || live_mul[i] % TM != 0)) return CNERF_E_ARG;
  }
  if (n == 1) a.net[1] = a.net[0];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CNERF_E_NODEVICE;
  int ns[MAX_WG_JOBS];
  {
    const int cap[2] = {nsplit[0], n > 1 ? nsplit[1] : nsplit[0]};
    plan_ranges(a, nj, cap, ns);
  }
  int64_t slabs[MAX_WG_JOBS];
  int order[MAX_WG_JOBS];
  for (int i = 0; i < nj; ++i) {
    WgJob& j = a.job[i];
    slabs[i] = a.net[j.net].Mp / TM;
    j.nsplit = ns[i];
    j.chunk = (int)(cn_div_up(slabs[i], (int64_t)ns[i]) * TM);
    // a range's operand rows sit behind one buffer resource each: 32-bit byte offsets
    const NetGeom& gg = *g[j.net];
    if ((int64_t)j.chunk * (gg.s_rows > gg.g_rows ? gg.s_rows : gg.g_rows) * 4 >= (int64_t)0x7fffffff) return CNERF_E_UNSUPPORTED;
    if (j.nsplit > nsplit[j.net]) return CNERF_E_UNSUPPORTED;
    r.nsplit[r0[j.net] + j.tensor] = j.nsplit;
    if (j.bias_tensor >= 0) r.nsplit[r0[j.net] + j.bias_tensor] = j.nsplit;
  }
  // grid order: longest workgroups first, jobs back to back
  order_jobs(a.job, nj, ns, slabs, order);
  WgArgs b = a;
  int first = 0;
  for (int oi = 0; oi < nj; ++oi) {
    b.job[oi] = a.job[order[oi]];
    b.job[oi].first = first;
    first += b.job[oi].nsplit;
  }
  b.nj = nj;
  const size_t lds_bytes = LDS_BYTES;
  // the 160 KiB dynamic-LDS opt-in is a per-device function attribute: set it once per device this process launches on
  // (idempotent, so a race between two host threads only repeats the call)
  static bool attr_set[2][64] = {};
  const void* kfn = bf3 ? reinterpret_cast<const void*>(wgrad_mixed_k) : reinterpret_cast<const void*>(wgrad_k);
  if (!attr_set[bf3 ? 1 : 0][dev]) {
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return (int)hipGetLastError();
    attr_set[bf3 ? 1 : 0][dev] = true;
  }
#ifdef CN_WGRAD_DYN
  static int* tickets[64] = {};
  static int ncu[64] = {};
  if (!tickets[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&tickets[dev]), 256) != hipSuccess) return (int)hipGetLastError();
    if (hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipGetLastError();
    if (const char* e = getenv("CNERF_WGRAD_DYN_GRID")) ncu[dev] = atoi(e);
  }
  if (hipMemsetAsync(tickets[dev], 0, 4, st) != hipSuccess) return (int)hipGetLastError();
  b.total = first;
  b.counter = tickets[dev];
  const int grid = first < ncu[dev] ? first : ncu[dev];
#else
  b.total = first;
  b.counter = nullptr;
  const int grid = first;
#endif
  if (bf3) hipLaunchKernelGGL(wgrad_mixed_k, dim3(grid), dim3(64 * NWAVES), lds_bytes, st, b);
  else hipLaunchKernelGGL(wgrad_k, dim3(grid), dim3(64 * NWAVES), lds_bytes, st, b);
  CN_CHECK_LAUNCH();
  r.accumulate = accumulate;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(64, nr), dim3(256), 0, st, r);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// Host-only view of the range plan (no launch, no device): for the GEMM jobs of one or two networks at Mp0 / Mp1 points ->
// per job {net, N, K, tiles of the busiest wave, ranges, points per range, destination tensor} (7 ints each, grid order).
// Returns the number of jobs.  Used by tests/test_host.py and scripts/wgrad_plan.py.
extern "C" int cnerf_debug_wgrad_plan(const cnerf_net* net0, int64_t Mp0, const cnerf_net* net1, int64_t Mp1, int* out7,
                                      int max_jobs) {
  NetGeom g[2];
  const cnerf_net* nets[2] = {net0, net1};
  const int64_t Mps[2] = {cn_round_up(Mp0, 32), cn_round_up(Mp1, 32)};
  const int n = net1 ? 2 : 1;
  WgArgs a;
  RedArgs r;
  cnerf_ptrs dummy;
  memset(&dummy, 0, sizeof(dummy));
  int nj = 0, nr = 0, cap[2] = {1, 1};
  for (int i = 0; i < n; ++i) {
    int rc = cn_make_geom(nets[i], &g[i]);
    if (rc) return rc < 0 ? rc : -rc;
    if (!add_net_jobs(g[i], i, nullptr, nullptr, Mps[i], nullptr, &dummy, a, nj, r, nr)) return CNERF_E_UNSUPPORTED;
    cap[i] = cn_wgrad_nsplit(Mps[i]);
  }
  if (n == 1) cap[1] = cap[0];
  if (nj > max_jobs) return CNERF_E_ARG;
  int ns[MAX_WG_JOBS], order[MAX_WG_JOBS];
  int64_t slabs[MAX_WG_JOBS];
  plan_ranges(a, nj, cap, ns);
  for (int i = 0; i < nj; ++i) slabs[i] = a.net[a.job[i].net].Mp / TM;
  order_jobs(a.job, nj, ns, slabs, order);
  for (int oi = 0; oi < nj; ++oi) {
    const WgJob& j = a.job[order[oi]];
    int* o = out7 + 7 * oi;
    o[0] = j.net; o[1] = j.N - j.n_lo; o[2] = j.K; o[3] = j.an * j.ak; o[4] = ns[order[oi]];
    o[5] = (int)(cn_div_up(slabs[order[oi]], (int64_t)ns[order[oi]]) * TM); o[6] = j.tensor;
  }
  return nj;
}

int cn_wgrad_launch(const NetGeom& g, const float* stash, const float* G, int64_t M, int64_t Mp, float* partials,
                    int nsplit, const cnerf_ptrs* grads, int accumulate, hipStream_t st, int bf3) {
  (void)M;   // padding points [M, Mp) are stored as zeros by the producers: no masking here
  const NetGeom* gp = &g;
  return cn_wgrad_launch_n(1, &gp, &stash, &G, &Mp, &partials, &nsplit, &grads, accumulate, st, bf3, nullptr, nullptr, nullptr);
}
