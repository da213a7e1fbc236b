// Weight gradients of the NeRF MLP: dW[n][k] = sum_m X^T[n][m] * Y^T[k][m]  (m = points)
// with X^T = gradient w.r.t. a layer's pre-activation and Y^T = that layer's input, both stored
// feature-major / point-contiguous ([rows][Mp]) by mlp_fwd / mlp_dgrad.  "NT" GEMMs on
// v_mfma_f32_32x32x2_f32 whose contraction runs over up to ~10^6 points.
//
// A workgroup (4 waves, one per SIMD) owns one GEMM's whole (<=256 x <=256) output and one of `nsplit` point
// ranges.  The 4 waves tile the output as a gn x gk grid chosen per GEMM so that all four have work
// (256x256 -> 2x2 waves of 4x4 MFMA tiles; 128x256 -> 1x4 of 4x2; 256x64 -> 2x2 of 4x1; 1x256 -> 1x4 ...).
// 32-point slabs of X^T / Y^T are staged through a DOUBLE-BUFFERED LDS image (row stride 36 floats:
// conflict-free ds_read_b128): slab s+1 is written to the other buffer after the MFMAs of slab s, one barrier
// per slab, and the global loads of slab s+2 are in flight under the 256 MFMAs per wave of slab s+1.
// GEMMs are launched largest-first over many small point ranges so the tail of the grid is short; split
// partials are reduced in a fixed order by a second kernel (bit-reproducible run to run).
#include "mlp_common.hpp"

namespace {

constexpr int MAX_WG_JOBS = 48;
constexpr int TM = 32;         // points per LDS slab
constexpr int LDR = TM + 4;    // padded LDS row (floats)
constexpr int TILE_FLOATS = 256 * LDR;

struct WgJob {
  int xrow, yrow;     // first row of X^T in G, of Y^T in the stash
  int N, K;           // valid output rows / cols
  int tensor;         // destination parameter tensor
  int ld, col0;       // its row stride and first column
  int bias_tensor;    // -1: none
  int gk, an, ak;     // wave grid: wave w -> (wn, wk) = (w / gk, w % gk) owns an x ak tiles of 32x32
};

struct WgArgs {
  WgJob job[MAX_WG_JOBS];
  int64_t toff[CNERF_MAX_TENSORS];   // offset of each tensor in the partial (parameter-space) buffer
  const float* stash;
  const float* G;
  float* partials;
  int64_t M, Mp, pstride;            // pstride = floats per split slice
  int64_t chunk;                     // points per split (multiple of 32)
};

__global__ __launch_bounds__(256) void wgrad_k(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 buffers][X tile | Y tile]
  const WgJob jb = a.job[blockIdx.y];
  const int split = blockIdx.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, i31 = lane & 31, hh = lane >> 5;
  const int wn = wv / jb.gk, wk = wv - wn * jb.gk;
  const int64_t m_begin = (int64_t)split * a.chunk;
  const int64_t m_end = m_begin + a.chunk < a.Mp ? m_begin + a.chunk : a.Mp;
  const float* X = a.G + (int64_t)jb.xrow * a.Mp;
  const float* Y = a.stash + (int64_t)jb.yrow * a.Mp;
  const int ntn = (jb.N + 31) >> 5, ntk = (jb.K + 31) >> 5;
  const int tn0 = wn * jb.an, tk0 = wk * jb.ak;   // first n / k tile of this wave
  f32x16 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // staging: thread -> (row = tid/8 + 32 i, 16-byte chunk = tid%8), i = 0..7, for X and for Y
  const int srow = tid >> 3, sch = tid & 7;
  const int nrx = ntn * 32, nry = ntk * 32;       // rows that are ever read back
  f32x4 px[8], py[8];
  auto fetch = [&](int64_t m0) {
    const int64_t col = m0 + 4 * sch;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 32 * i;
      f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vy = {0.f, 0.f, 0.f, 0.f};
      if (r < jb.N) vx = *reinterpret_cast<const f32x4*>(X + (int64_t)r * a.Mp + col);
      if (r < jb.K) vy = *reinterpret_cast<const f32x4*>(Y + (int64_t)r * a.Mp + col);
      if (col + 3 >= a.M) {   // padding columns [M, Mp) hold garbage: zero them on both sides
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j >= a.M) { vx[j] = 0.f; vy[j] = 0.f; }
      }
      px[i] = vx; py[i] = vy;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 32 * i;
      if (r < nrx) *reinterpret_cast<f32x4*>(buf + r * LDR + 4 * sch) = px[i];
      if (r < nry) *reinterpret_cast<f32x4*>(buf + TILE_FLOATS + r * LDR + 4 * sch) = py[i];
    }
  };

  int cur = 0;
  if (m_begin < m_end) {
    fetch(m_begin);
    commit(lds);
    __syncthreads();
    if (m_begin + TM < m_end) fetch(m_begin + TM);
  }
  for (int64_t m0 = m_begin; m0 < m_end; m0 += TM) {
    const float* Xs = lds + cur * 2 * TILE_FLOATS;
    const float* Ys = Xs + TILE_FLOATS;
#pragma unroll
    for (int st = 0; st < TM / 8; ++st) {
      f32x4 av[4], bv[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (x < jb.an) av[x] = *reinterpret_cast<const f32x4*>(Xs + (32 * (tn0 + x) + i31) * LDR + 8 * st + 4 * hh);
        if (x < jb.ak) bv[x] = *reinterpret_cast<const f32x4*>(Ys + (32 * (tk0 + x) + i31) * LDR + 8 * st + 4 * hh);
      }
      if (wk == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (x < jb.an) bsum[x] += (av[x][0] + av[x][1]) + (av[x][2] + av[x][3]);
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (x < jb.an && tn0 + x < ntn) {
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (y < jb.ak && tk0 + y < ntk) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[x][y] = mfma(av[x][j], bv[y][j], acc[x][y]);
            }
          }
        }
      }
    }
    if (m0 + TM < m_end) commit(lds + (cur ^ 1) * 2 * TILE_FLOATS);   // other buffer: its readers passed the last barrier
    __syncthreads();
    if (m0 + 2 * TM < m_end) fetch(m0 + 2 * TM);
    cur ^= 1;
  }

  float* out = a.partials + (int64_t)split * a.pstride;
  float* Wout = out + a.toff[jb.tensor];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (x >= jb.an || y >= jb.ak) continue;
      const int k = 32 * (tk0 + y) + i31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * (tn0 + x) + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < jb.N && k < jb.K) Wout[(int64_t)n * jb.ld + jb.col0 + k] = acc[x][y][r];
      }
    }
  if (jb.bias_tensor >= 0 && wk == 0) {
    float* Bout = out + a.toff[jb.bias_tensor];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (x >= jb.an) continue;
      const float s = bsum[x] + __shfl_xor(bsum[x], 32, 64);
      const int n = 32 * (tn0 + x) + i31;
      if (hh == 0 && n < jb.N) Bout[n] = s;
    }
  }
}

struct RedArgs {
  float* grad[CNERF_MAX_TENSORS];
  int64_t toff[CNERF_MAX_TENSORS];
  int64_t numel[CNERF_MAX_TENSORS];
  int touched[CNERF_MAX_TENSORS];
  const float* partials;
  int64_t pstride;
  int nsplit, accumulate;
};

__global__ void wgrad_reduce_k(RedArgs a) {
  const int t = blockIdx.y;
  float* g = a.grad[t];
  if (g == nullptr) return;
  const int64_t n = a.numel[t];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    if (a.touched[t]) {
      const float* p = a.partials + a.toff[t] + i;
      for (int k = 0; k < a.nsplit; ++k) s += p[(int64_t)k * a.pstride];
    }
    g[i] = a.accumulate ? g[i] + s : s;
  }
}

}  // namespace

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
  // many small point ranges (>= 64 slabs of 32 points each) so that the ~15 GEMMs x nsplit workgroups of
  // unequal size pack well onto 256 CUs; capped to bound the partial-gradient buffer (nsplit x 4.8 MB)
  int64_t s = Mp / 2048;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

int cn_wgrad_launch(const NetGeom& g, const float* stash, const float* G, int64_t M, int64_t Mp, float* partials,
                    int nsplit, const cnerf_ptrs* grads, int accumulate, hipStream_t st) {
  cnerf_net net{g.D, g.W, g.L, g.Ld, g.viewdirs, g.out_ch, g.skip};
  WgArgs a;
  RedArgs r;
  const int nt = cnerf_num_tensors(&net);
  int64_t off = 0;
  for (int i = 0; i < nt; ++i) {
    int64_t rr, cc;
    cnerf_tensor_shape(&net, i, &rr, &cc);
    a.toff[i] = r.toff[i] = off;
    r.numel[i] = rr * cc;
    r.grad[i] = grads->p[i];
    r.touched[i] = 0;
    off += cn_round_up(rr * cc, 4);
  }
  const int64_t pstride = cn_round_up(off, 64);
  int nj = 0;
  const int D = g.D, W = g.W, Wh = g.Wh;
  auto add = [&](int xrow, int yrow, int N, int K, int tensor, int ld, int col0, int bias_tensor) {
    const int ntn = (N + 31) / 32, ntk = (K + 31) / 32;
    // wave grid gn x gk in {1x4, 2x2, 4x1}: an x ak <= 4x4 tiles per wave; minimise the busiest wave's tile
    // count (= the workgroup's MFMA time), then the operand traffic an+ak
    int best_gk = 0, best_an = 0, best_ak = 0, best_cost = 1 << 30;
    for (int gn = 1; gn <= 4; gn *= 2) {
      const int gk = 4 / gn;
      const int an = (ntn + gn - 1) / gn, ak = (ntk + gk - 1) / gk;
      if (an > 4 || ak > 4) continue;
      const int cost = an * ak * 16 + an + ak;
      if (cost < best_cost) { best_cost = cost; best_gk = gk; best_an = an; best_ak = ak; }
    }
    a.job[nj++] = WgJob{xrow, yrow, N, K, tensor, ld, col0, bias_tensor, best_gk ? best_gk : 2, best_an, best_ak};
    r.touched[tensor] = 1;
    if (bias_tensor >= 0) r.touched[bias_tensor] = 1;
  };
  const int base = 2 * D;
  // largest GEMMs first (the grid is dispatched job-major): the small ones fill the tail
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
    add(g.g_out + 3, g.s_h[D - 1], 1, W, base + 4, W, 0, base + 5);
    add(g.g_out, g.s_hv, 3, Wh, base + 6, Wh, 0, base + 7);
  } else {
    add(g.g_out, g.s_h[D - 1], g.out_ch, W, base + 2, W, 0, base + 3);
  }
  if (nj > MAX_WG_JOBS) return CNERF_E_UNSUPPORTED;
  for (int j = 0; j < nj; ++j) {   // every tile must be owned by a wave
    const WgJob& jb = a.job[j];
    const int gn = 4 / jb.gk;
    if (gn * jb.an * 32 < jb.N || jb.gk * jb.ak * 32 < jb.K) return CNERF_E_UNSUPPORTED;
  }
  a.stash = stash; a.G = G; a.partials = partials; a.M = M; a.Mp = Mp; a.pstride = pstride;
  a.chunk = cn_round_up(cn_div_up(Mp, nsplit), TM);
  const size_t lds_bytes = (size_t)4 * TILE_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_k, dim3(nsplit, nj), dim3(256), lds_bytes, st, a);
  CN_CHECK_LAUNCH();
  r.partials = partials; r.pstride = pstride; r.nsplit = nsplit; r.accumulate = accumulate;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(32, nt), dim3(256), 0, st, r);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
