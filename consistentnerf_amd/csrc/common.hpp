// Shared helpers for the gfx950 kernels behind include/cnerf.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cnerf.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CN_CHECK_LAUNCH()                               \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return (int)e__;             \
  } while (0)

static inline hipStream_t cn_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cn_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t cn_round_up(int64_t a, int64_t b) { return cn_div_up(a, b) * b; }

// wave64 helpers (one wavefront = 64 lanes on CDNA4)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- network geometry shared by pack / fwd / bwd -----------------------------------------------
// Everything is expressed in "panels": a linear map y[N] = W[N,K] x[K] is stored for the MFMA A-operand
// as P[K/8][Np][8] floats, P[kg][n][j] = W[n][8*kg + j] (zero padded), so that lane (i = lane&31,
// hh = lane>>5) of a wave fetches the 4 consecutive k it feeds to 4 successive
// v_mfma_f32_32x32x2_f32 with ONE 16-byte load at P + ((kg*Np + n0 + i)*8 + 4*hh), and a wave's 64
// lanes read 1 KiB contiguous.  Forward panels of biased layers carry ONE MORE group, P[K/8][n][0] = bias[n]:
// the bias is added by a last MFMA step against a constant B operand (1 for the hh=0 half-wave), so no bias
// vector is ever loaded into registers.
struct NetGeom {
  int D, W, NT;             // NT = W/32 output tiles of a W-wide layer
  int in_ch, in_chp;        // gamma(x) channels (63) and padded to a multiple of 32 (64)
  int dir_ch, dir_chp;      // gamma(d) channels (27) / padded (32); 0 without viewdirs
  int L, Ld;                // encoding frequencies
  int viewdirs, out_ch, skip;
  int Wh;                   // W/2 (view branch width)
  // offsets (floats) into the packed buffer
  int64_t f_l0;             // [in_chp/8][W][8]
  int64_t f_trunk[16];      // l=1..D-1: [W/8][W][8]
  int64_t f_skip;           // [in_chp/8][W][8] (gamma(x) columns of layer skip+1)
  int64_t f_feat;           // [W/8][W][8]
  int64_t f_views;          // [W/8][Wh][8]
  int64_t f_viewsd;         // [dir_chp/8][Wh][8]
  int64_t t_trunk[16];      // transposed panels for dgrad, l=1..D-1: [W/8][W][8] over (n-groups, k rows)
  int64_t t_feat;           // [W/8][W][8]
  int64_t t_views;          // [Wh/8][W][8]
  int64_t v_alpha;          // [W]
  int64_t v_rgb;            // [3][Wh]
  int64_t v_out;            // [out_ch][W]   (no-viewdirs head)
  int64_t b_trunk[16];      // biases [W] each, l=0..D-1
  int64_t b_feat, b_views, b_alpha, b_rgb, b_out;
  int64_t total;
  // stash (training): logically [Mp][s_rows], Mp = M rounded up to 32, column offsets of each block below; stored
  // TILE-MAJOR: tiles of 32 points x 8 columns (1 KiB, point-major inside), element (p, c) at float index
  // ((p/32 * s_rows/8 + c/8) * 32 + p%32) * 8 + c%8 — the 64 lanes of a producing wave write one tile with one
  // 16-byte store each, 1 KiB contiguous, and the wgrad DMA moves a tile with one instruction.
  // s_mask: ReLU sign bits of every hidden layer (1 = pre-activation > 0), packed per lane of the producing wave:
  // layer block b (trunk l = 0..D-1, then the view branch) starts at s_mask + s_mb[b]; inside it half-wave hh owns
  // md dwords (md = tiles/2 rounded up), dword d = tiles 2d, 2d+1: upper 16 bits = even registers r of tile 2d then of
  // tile 2d+1 (first = MSB), lower 16 bits = odd registers likewise; a single-tile dword holds 8 + 8 bits left-aligned
  // (mlp_common.hpp relu_bits).  The backward reads masks from here, not from the activations.
  int s_enc, s_h[16], s_feat, s_denc, s_hv, s_mask, s_mb[17], s_rows;
  // backward workspace (gradient wrt pre-activations), logically [Mp][g_rows], same tile-major storage
  int g_z[16], g_feat, g_hv, g_out, g_rows;
};

int cn_make_geom(const cnerf_net* net, NetGeom* g);

// Rays (= waves) per workgroup of the loss-folding compositing launches (composite.hip) — and therefore the unit of their
// per-workgroup partial sums, which the loss tails (loss.hip) index: ONE definition for both files.
constexpr int CN_CLOSS_RAYS_PER_WG = 8;

