// Weight gradients of the NeRF MLP: dW[n][k] = sum_m X^T[n][m] * Y^T[k][m]  (m = points)
// with X^T = gradient w.r.t. a layer's pre-activation and Y^T = that layer's input, both stored
// feature-major / point-contiguous ([rows][Mp]) by mlp_fwd / mlp_dgrad.  "NT" GEMMs on
// v_mfma_f32_32x32x2_f32 whose contraction runs over up to ~10^6 points: a workgroup (4 waves) owns a
// 256x256 output tile (each wave 128x128 = 4x4 MFMA tiles, 256 accumulators) and one of `nsplit` point
// ranges; 32-point slabs of X^T and Y^T are staged through LDS (row stride 36 floats: conflict-free
// ds_read_b128), the next slab's global loads are in flight under the current slab's 256 MFMAs per wave.
// Split partials are reduced in a fixed order by a second kernel (bit-reproducible run to run).
#include "mlp_common.hpp"

namespace {

constexpr int MAX_WG_JOBS = 48;
constexpr int TM = 32;         // points per LDS slab
constexpr int LDR = TM + 4;    // padded LDS row (floats)

struct WgJob {
  int xrow, yrow;     // first row of X^T in G, of Y^T in the stash
  int N, K;           // valid output rows / cols
  int tensor;         // destination parameter tensor
  int ld, col0;       // its row stride and first column
  int bias_tensor;    // -1: none
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
  __shared__ __attribute__((aligned(16))) float Xs[256 * LDR];
  __shared__ __attribute__((aligned(16))) float Ys[256 * LDR];
  const WgJob jb = a.job[blockIdx.y];
  const int split = blockIdx.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, i31 = lane & 31, hh = lane >> 5;
  const int wn = wv >> 1, wk = wv & 1;
  const int64_t m_begin = (int64_t)split * a.chunk;
  const int64_t m_end = m_begin + a.chunk < a.Mp ? m_begin + a.chunk : a.Mp;
  const float* X = a.G + (int64_t)jb.xrow * a.Mp;
  const float* Y = a.stash + (int64_t)jb.yrow * a.Mp;
  // n/k tiles (of 32) this wave owns: [4wn, 4wn+4) x [4wk, 4wk+4), clipped to the job
  const int ntn = (jb.N + 31) >> 5, ntk = (jb.K + 31) >> 5;
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
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 32 * i;
      *reinterpret_cast<f32x4*>(Xs + r * LDR + 4 * sch) = px[i];
      *reinterpret_cast<f32x4*>(Ys + r * LDR + 4 * sch) = py[i];
    }
  };

  if (m_begin < m_end) fetch(m_begin);
  for (int64_t m0 = m_begin; m0 < m_end; m0 += TM) {
    commit();
    __syncthreads();
    if (m0 + TM < m_end) fetch(m0 + TM);
#pragma unroll
    for (int st = 0; st < TM / 8; ++st) {
      f32x4 av[4], bv[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        av[x] = *reinterpret_cast<const f32x4*>(Xs + (32 * (4 * wn + x) + i31) * LDR + 8 * st + 4 * hh);
        bv[x] = *reinterpret_cast<const f32x4*>(Ys + (32 * (4 * wk + x) + i31) * LDR + 8 * st + 4 * hh);
      }
      if (wk == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x) bsum[x] += (av[x][0] + av[x][1]) + (av[x][2] + av[x][3]);
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (4 * wn + x < ntn) {
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (4 * wk + y < ntk) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[x][y] = mfma(av[x][j], bv[y][j], acc[x][y]);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  float* out = a.partials + (int64_t)split * a.pstride;
  float* Wout = out + a.toff[jb.tensor];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int k = 32 * (4 * wk + y) + i31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * (4 * wn + x) + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < jb.N && k < jb.K) Wout[(int64_t)n * jb.ld + jb.col0 + k] = acc[x][y][r];
      }
    }
  if (jb.bias_tensor >= 0 && wk == 0) {
    float* Bout = out + a.toff[jb.bias_tensor];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float s = bsum[x] + __shfl_xor(bsum[x], 32, 64);
      const int n = 32 * (4 * wn + x) + i31;
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
  // ~32 point ranges (x ~15 output tiles per net -> ~2 workgroups per CU), never finer than one slab
  int64_t s = Mp / 1024;
  if (s < 1) s = 1;
  if (s > 32) s = 32;
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
    a.job[nj++] = WgJob{xrow, yrow, N, K, tensor, ld, col0, bias_tensor};
    r.touched[tensor] = 1;
    if (bias_tensor >= 0) r.touched[bias_tensor] = 1;
  };
  add(g.g_z[0], g.s_enc, W, g.in_ch, 0, g.in_ch, 0, 1);
  for (int l = 1; l < D; ++l) {
    const bool sk = g.skip >= 0 && l == g.skip + 1;
    const int ld = sk ? W + g.in_ch : W;
    add(g.g_z[l], g.s_h[l - 1], W, W, 2 * l, ld, sk ? g.in_ch : 0, 2 * l + 1);
    if (sk) add(g.g_z[l], g.s_enc, W, g.in_ch, 2 * l, ld, 0, -1);
  }
  const int base = 2 * D;
  if (g.viewdirs) {
    const int ldv = W + g.dir_ch;
    add(g.g_hv, g.s_feat, Wh, W, base + 0, ldv, 0, base + 1);
    add(g.g_hv, g.s_denc, Wh, g.dir_ch, base + 0, ldv, W, -1);
    add(g.g_feat, g.s_h[D - 1], W, W, base + 2, W, 0, base + 3);
    add(g.g_out + 3, g.s_h[D - 1], 1, W, base + 4, W, 0, base + 5);
    add(g.g_out, g.s_hv, 3, Wh, base + 6, Wh, 0, base + 7);
  } else {
    add(g.g_out, g.s_h[D - 1], g.out_ch, W, base + 2, W, 0, base + 3);
  }
  if (nj > MAX_WG_JOBS) return CNERF_E_UNSUPPORTED;
  a.stash = stash; a.G = G; a.partials = partials; a.M = M; a.Mp = Mp; a.pstride = pstride;
  a.chunk = cn_round_up(cn_div_up(Mp, nsplit), TM);
  hipLaunchKernelGGL(wgrad_k, dim3(nsplit, nj), dim3(256), 0, st, a);
  CN_CHECK_LAUNCH();
  r.partials = partials; r.pstride = pstride; r.nsplit = nsplit; r.accumulate = accumulate;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(32, nt), dim3(256), 0, st, r);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
