// Masked photometric / depth losses of ConsistentNeRF (V:1645-1648, V:1737, V:1786-1788, V:1865) and the
// vanilla MSE (R:769), forward value + gradient seeds for the compositing backward in one launch.
// Replaces 4 boolean-index gathers (each a host sync) per level.  One workgroup, fixed reduction order.
#include "common.hpp"

namespace {

constexpr int T = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < T / 64; ++i) s += sh[i];
  return s;
}

__global__ __launch_bounds__(T) void masked_loss_k(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                   const float* __restrict__ depth, const float* __restrict__ prior,
                                                   const float* __restrict__ mask, int64_t B, float far, float coef,
                                                   const float* __restrict__ counts, float g_scale,
                                                   float* __restrict__ loss, float* __restrict__ d_rgb,
                                                   float* __restrict__ d_depth) {
  __shared__ double sh[T / 64];
  // pass 1: counts and squared-error sums of the two sets (m==1, m==0; other values belong to neither)
  double n1 = 0, n0 = 0, s1 = 0, s0 = 0, sd = 0;
  for (int64_t i = threadIdx.x; i < B; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const bool in1 = m == 1.f, in0 = m == 0.f;
    float e = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = rgb[3 * i + c] - tgt[3 * i + c];
      e += d * d;
    }
    if (in1) { n1 += 1.0; s1 += (double)e; }
    if (in0) { n0 += 1.0; s0 += (double)e; }
    if (depth && in1) {
      const float d = depth[i] / far - prior[i] / far;
      sd += (double)(d * d);
    }
  }
  n1 = block_sum(n1, sh); n0 = block_sum(n0, sh);
  s1 = block_sum(s1, sh); s0 = block_sum(s0, sh); sd = block_sum(sd, sh);
  // global counts (e.g. all-reduced over ranks) override the local ones for the normalisation
  const double N1 = counts ? (double)counts[0] : n1;
  const double N0 = counts ? (double)counts[1] : n0;
  if (threadIdx.x == 0) {
    float l = (float)(s1 / (3.0 * N1));
    if (N0 > 0) l += coef * (float)(s0 / (3.0 * N0));
    loss[0] = l;
    loss[1] = depth ? (float)(sd / N1) : 0.f;
  }
  const float w1 = g_scale * (float)(2.0 / (3.0 * N1));
  const float w0 = N0 > 0 ? g_scale * coef * (float)(2.0 / (3.0 * N0)) : 0.f;
  const float wd = g_scale * (float)(2.0 / N1) / far;
  for (int64_t i = threadIdx.x; i < B; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const float w = m == 1.f ? w1 : (m == 0.f ? w0 : 0.f);
    if (d_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d_rgb[3 * i + c] = w * (rgb[3 * i + c] - tgt[3 * i + c]);
    }
    if (d_depth) d_depth[i] = (depth && m == 1.f) ? wd * (depth[i] / far - prior[i] / far) : 0.f;
  }
}


// ---- f-5: monocular-depth patch term (V:1678-1720) -------------------------------------------------------------------
// Per patch of n rays: inverse rendered depth pr = nan_to_num(1 / where(d <= 0, 1e-4, d)) and the monocular prior
// gt = nan_to_num(mono) are each min-max normalised over the prior's valid set m = (gt > 0), aligned by the mean
// difference, and the mean squared residual / P / 2 is summed over the P patches.  One wave64 per patch (n = 256 ->
// 4 elements per lane), everything re-read from L1/L2 between the reduction passes; butterfly reductions in a fixed
// order, the P partial losses summed by thread 0.  The backward follows autograd's rules for the same expression:
// full-reduction min()/max() split their gradient evenly over ties, nan_to_num passes it where 1/x is finite,
// d(1/x) = -g (1/x)^2, and nothing flows where d <= 0 (the constant branch of the where()).
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float nan_to_num(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.402823466e+38f;
  if (v == -INFINITY) return -3.402823466e+38f;
  return v;
}

// one patch by one wave64: returns (all lanes) the patch's share of the loss, (float)(mean residual^2) / P / 2; writes d_depth[n]
// (the gradient of that share times g_scale) when asked
__device__ __forceinline__ float patch_wave(const float* __restrict__ d, const float* __restrict__ g, int P, int n, float g_scale,
                                            float* __restrict__ d_depth, int lane) {
  auto inv = [&](int i) { const float x = d[i] <= 0.f ? 1e-4f : d[i]; return 1.f / x; };
  // pass 1: ranges
  float gmin = 1e5f, gmax = -INFINITY, pmin = 1e5f, pmax = -INFINITY;
  for (int i = lane; i < n; i += 64) {
    const float gt = nan_to_num(g[i]), pr = nan_to_num(inv(i));
    const float m = gt > 0.f ? 1.f : 0.f;
    gmin = fminf(gmin, gt > 0.f ? gt : 1e5f);
    gmax = fmaxf(gmax, gt);
    pmin = fminf(pmin, m * pr > 0.f ? pr : 1e5f);
    pmax = fmaxf(pmax, m * pr);
  }
  gmin = wave_min(gmin); gmax = wave_max(gmax); pmin = wave_min(pmin); pmax = wave_max(pmax);
  const float rg = gmax - gmin + 1e-4f, r = pmax - pmin + 1e-4f;
  auto gtn = [&](int i) { const float gt = nan_to_num(g[i]); return (gt > 0.f ? 1.f : 0.f) * (gt - gmin) / rg; };
  auto prq = [&](int i) { const float gt = nan_to_num(g[i]); return (gt > 0.f ? 1.f : 0.f) * (nan_to_num(inv(i)) - pmin); };
  // pass 2: shift
  double sd = 0.0;
  for (int i = lane; i < n; i += 64) sd += (double)(prq(i) / r - gtn(i));
  const float alpha = (float)(wave_sum(sd) / n);
  // pass 3: residuals
  double se2 = 0.0, se = 0.0;
  for (int i = lane; i < n; i += 64) {
    const float e = gtn(i) - prq(i) / r + alpha;
    se2 += (double)(e * e); se += (double)e;
  }
  se2 = wave_sum(se2); se = wave_sum(se);
  const float share = (float)(se2 / n) / P / 2;
  if (!d_depth) return share;
  // pass 4: gradient sums.  dL/dprn_i = (mean(e) - e_i) / (n P)
  const float ebar = (float)(se / n), w = g_scale / ((float)n * (float)P);
  double s_mg = 0.0, s_gq = 0.0;
  int cmin = 0, cmax = 0;
  for (int i = lane; i < n; i += 64) {
    const float gt = nan_to_num(g[i]), pr = nan_to_num(inv(i));
    const float m = gt > 0.f ? 1.f : 0.f, q = m * (pr - pmin);
    const float gi = w * (ebar - (gtn(i) - q / r + alpha));
    s_mg += (double)(m * gi); s_gq += (double)(gi * q);
    cmin += ((m * pr > 0.f ? pr : 1e5f) == pmin);
    cmax += (m * pr == pmax);
  }
  s_mg = wave_sum(s_mg); s_gq = wave_sum(s_gq);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cmin += __shfl_xor(cmin, o, 64); cmax += __shfl_xor(cmax, o, 64); }
  const float d_r = (float)(-s_gq / ((double)r * (double)r));          // dL/dr
  const float d_pmin = (float)(-s_mg / (double)r) - d_r, d_pmax = d_r;    // r = pmax - pmin + 1e-4
  for (int i = lane; i < n; i += 64) {
    const float gt = nan_to_num(g[i]), c = inv(i), pr = nan_to_num(c);
    const float m = gt > 0.f ? 1.f : 0.f, q = m * (pr - pmin);
    const float gi = w * (ebar - (gtn(i) - q / r + alpha));
    float dpr = m * gi / r;
    if (m * pr > 0.f && pr == pmin) dpr += d_pmin / (float)cmin;
    if (m * pr == pmax) dpr += m * d_pmax / (float)cmax;
    const float dc = (c == c && fabsf(c) != INFINITY) ? dpr : 0.f;     // nan_to_num backward
    const float dx = -dc * c * c;                                      // reciprocal backward
    d_depth[i] = d[i] <= 0.f ? 0.f : dx;
  }
  return share;
}

__global__ void patch_depth_loss_k(const float* __restrict__ depth, const float* __restrict__ mono, int P, int n,
                                   float g_scale, float* __restrict__ loss, float* __restrict__ d_depth) {
  __shared__ float part[16];
  const int p = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float share = patch_wave(depth + (int64_t)p * n, mono + (int64_t)p * n, P, n, g_scale,
                                 d_depth ? d_depth + (int64_t)p * n : nullptr, lane);
  if (lane == 0) part[p] = share;
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    for (int k = 0; k < P; ++k) l += part[k];
    loss[0] = l;
  }
}

// ---- the loss tail of a ConsistentNeRF step whose masked losses rode in the compositing launches (composite.hip ClossFwd) -------
// ONE workgroup: sums each level's five per-workgroup fp64 partials in index order (thread t takes entries t, t + T, ...; then the
// fixed-order block sum — the value does not depend on the grid's scheduling), normalises with the local or the caller's (global,
// sharded batch) counts exactly as masked_loss_k does, evaluates the monocular patch term of both levels (one wave per patch: waves
// [0, P) the last level, [P, 2P) the coarse one), assembles the step's loss in the reference's order of accumulation (V:1672-1865:
// loss += w_rgb img_loss; loss += w_patch mono_mse; loss += w_depth depth_loss; then the same three of the coarse level) and leaves
// the per-level seed weights (w1, w0, wd) for cnerf_composite_bwd_closs.
struct ClossTail {
  const double* part[2];     // [5][nparts] per level; level 0 = the last (fine) level, level 1 = coarse or nullptr
  int nparts;
  const float* counts;       // (n1, n0) or nullptr
  float coef, far, rgb_w, depth_w, patch_w;
  int has_depth;
  const float* depth[2];     // depth maps of the levels (patch term), or nullptr
  const float* mono;
  int P, n;
  float* patch_d[2];         // [P * n] per level or nullptr
  float* terms;              // [8]: loss, img_loss, depth_loss, patch_loss, img_loss0, depth_loss0, patch_loss0, -
  float* stats;              // [2][4]: w1, w0, wd, - per level
  int ss;                    // 1: the in-loop consistency step's primary terms (VT:941-969), per-term coins below;
                             // 2: the WHOLE step as one render (cnerf_closs_finish_ss2): partials [0, nparts1) = the primary rays,
                             //    [nparts1, nparts) = the warped rays of the second render (VT:927-938) + padding
  int coin[4];               // rgb, depth, rgb0, depth0: 1 = the term over the selected rays (mask == 1), 0 = the reference's other branch
  int nparts1;               // ss == 2: workgroups of the first segment
  const float* counts3;      // ss == 2: GLOBAL (selected primary rays, primary rays, warped rays) of a batch sharded over ranks, or nullptr
};

__global__ __launch_bounds__(T) void closs_tail_k(ClossTail a) {
  __shared__ double sh[T / 64];
  __shared__ float pshare[2][8];
  __shared__ double tot[2][5];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int levels = a.part[1] ? 2 : 1;
  for (int lv = 0; lv < levels; ++lv)
    for (int k = 0; k < 5; ++k) {
      double s = 0.0;
      for (int i = threadIdx.x; i < a.nparts; i += T) s += a.part[lv][(int64_t)k * a.nparts + i];
      s = block_sum(s, sh);
      if (threadIdx.x == 0) tot[lv][k] = s;
    }
  if (a.P > 0 && wv < levels * a.P) {
    const int lv = wv / a.P, p = wv - lv * a.P;
    const float share = patch_wave(a.depth[lv] + (int64_t)p * a.n, a.mono + (int64_t)p * a.n, a.P, a.n, 1.f,
                                   a.patch_d[lv] ? a.patch_d[lv] + (int64_t)p * a.n : nullptr, lane);
    if (lane == 0) pshare[lv][p] = share;
  }
  if (a.ss == 2) {
    // The one-render form of the whole `--ss_loss` step (VT:899-969): rays [0, 8 nparts1) are the primary batch (mask = sel, prior =
    // depth_cas_s), rays from there on the second render's warped rays (mask = 1 on the live rows, 0 on the padding; target / prior =
    // the reference view's colours / depth prior at the snapped pixels).  Segment sums in index order, like the sums above.
    __shared__ double seg[2][2][5];
    for (int lv = 0; lv < levels; ++lv)
      for (int sg = 0; sg < 2; ++sg)
        for (int k = 0; k < 5; ++k) {
          const int lo = sg ? a.nparts1 : 0, hi = sg ? a.nparts : a.nparts1;
          double s = 0.0;
          for (int i = lo + (int)threadIdx.x; i < hi; i += T) s += a.part[lv][(int64_t)k * a.nparts + i];
          s = block_sum(s, sh);
          if (threadIdx.x == 0) seg[lv][sg][k] = s;
        }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // second render first, as the reference accumulates (VT:930-938): img2mse(rgb_ref, target_ref) [+ img2mse(depth_ref, prior_ref)]
    // then the coarse level's two; then the primary render's four coin-gated terms (VT:941-969, see the ss == 1 branch below)
    const double M = a.counts3 ? (double)a.counts3[2] : seg[0][1][3];
    float loss = 0.f;
    float ref_t[4] = {0.f, 0.f, 0.f, 0.f};
    // M == 0 (no point of the batch projects into the reference view; the reference's `while mask.sum() == 0` never ends there, and the
    // two-render route raises): the second render has no rays — its terms and seed weights are 0 instead of 0 / 0, terms[7] = 0 tells
    // the caller, and with no in-bounds ray `sel` is empty too: the primary terms below fall to their un-masked branch.
    const bool have2 = M > 0.0;
    for (int lv = 0; lv < levels; ++lv) {          // level 0 = the last (fine) level: its terms come first in the reference
      const float il = have2 ? (float)(seg[lv][1][0] / (3.0 * M)) : 0.f;
      loss = loss + il;
      ref_t[2 * lv] = il;
      float dl = 0.f;
      if (a.has_depth && have2) { dl = (float)(seg[lv][1][2] / M); loss = loss + dl; }
      ref_t[2 * lv + 1] = dl;
      float* st = a.stats + 8 * lv + 4;            // segment 2 of this level: live rows weigh 2 / (3 M) and 2 / M, padding rows 0
      st[0] = have2 ? (float)(2.0 / (3.0 * M)) : 0.f; st[1] = 0.f;
      st[2] = (a.has_depth && have2) ? (float)(2.0 / M) / a.far : 0.f; st[3] = 0.f;
    }
    const double s1 = seg[0][0][0], s0 = seg[0][0][1];
    const double N1 = a.counts3 ? (double)a.counts3[0] : seg[0][0][3];
    const double Nall = a.counts3 ? (double)a.counts3[1] : seg[0][0][3] + seg[0][0][4];
    const float plain = (float)((s1 + s0) / (3.0 * Nall));
    const bool anysel = N1 > 0.0;                 // (false only with M == 0, see above: the coins then have nothing to select)
    const bool c0 = a.coin[0] && anysel, c2 = a.coin[2] && anysel;
    const float wall = (float)(2.0 / (3.0 * Nall)), wsel = anysel ? (float)(2.0 / (3.0 * N1)) : 0.f;
    const float il = c0 ? (float)(s1 / (3.0 * N1)) : plain;
    float w1 = c0 ? wsel : wall, w0 = c0 ? 0.f : wall;
    loss = loss + il;
    float dl = 0.f;
    const bool dep = a.has_depth && a.coin[1] && anysel;
    if (dep) { dl = (float)(seg[0][0][2] / N1); loss = loss + dl; }
    a.terms[1] = il; a.terms[2] = dl; a.terms[3] = 0.f;
    a.stats[2] = dep ? (float)(2.0 / N1) / a.far : 0.f;
    a.stats[3] = 0.f;
    a.terms[4] = a.terms[5] = a.terms[6] = 0.f;
    if (levels == 2) {
      const float il0 = c2 ? (float)(seg[1][0][0] / (3.0 * N1)) : plain;
      loss = loss + il0;
      float dl0 = 0.f;
      const bool dep0 = a.has_depth && a.coin[3] && anysel;
      if (dep0) { dl0 = (float)(seg[1][0][2] / N1); loss = loss + dl0; }
      a.terms[4] = il0; a.terms[5] = dl0;
      a.stats[8] = c2 ? (float)(2.0 / (3.0 * N1)) : 0.f;
      a.stats[9] = 0.f;
      a.stats[10] = dep0 ? (float)(2.0 / N1) / a.far : 0.f;
      a.stats[11] = 0.f;
      if (!c2) { w1 += wall; w0 += wall; }
    }
    a.stats[0] = w1; a.stats[1] = w0;
    a.terms[0] = loss;
    a.terms[7] = (float)M;
    a.terms[8] = ref_t[0]; a.terms[9] = ref_t[1]; a.terms[10] = ref_t[2]; a.terms[11] = ref_t[3];
    return;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (a.ss) {
    // VT:941-969 — the primary render's terms under `--ss_loss`, mask = the rays `[mask_bound][mask]` selects (cnerf_ss_ref_rays'
    // `sel`), each term behind its own random.randint(0, 1) coin:
    //   img_loss  = coin ? img2mse(rgb[sel], target[sel])      : img2mse(rgb, target)                       (:942)
    //   depth     = coin ? img2mse(depth[sel], prior[sel])      : 0                     (un-normalised: far = 1)   (:950-952)
    //   img_loss0 = coin ? img2mse(rgb0[sel], target[sel])      : img2mse(rgb, target)  (the FINE rgb: the reference's line :959)
    //   depth0    = coin ? img2mse(depth0[sel], prior[sel])     : 0                                          (:966-968)
    // accumulated in that order (loss += each).  The fallback of :959 sends its gradient to the fine level's colours.
    const double s1 = tot[0][0], s0 = tot[0][1], N1 = tot[0][3], Nall = tot[0][3] + tot[0][4];
    const float plain = (float)((s1 + s0) / (3.0 * Nall));
    const float wall = (float)(2.0 / (3.0 * Nall)), wsel = (float)(2.0 / (3.0 * N1));
    const float il = a.coin[0] ? (float)(s1 / (3.0 * N1)) : plain;
    float w1 = a.coin[0] ? wsel : wall, w0 = a.coin[0] ? 0.f : wall;
    float loss = il, dl = 0.f;
    const bool dep = a.has_depth && a.coin[1];
    if (dep) { dl = (float)(tot[0][2] / N1); loss = loss + dl; }
    a.terms[1] = il; a.terms[2] = dl; a.terms[3] = 0.f;
    a.stats[2] = dep ? (float)(2.0 / N1) / a.far : 0.f;
    a.stats[3] = 0.f;
    a.terms[4] = a.terms[5] = a.terms[6] = 0.f;
    if (levels == 2) {
      const double N1c = tot[1][3];
      const float il0 = a.coin[2] ? (float)(tot[1][0] / (3.0 * N1c)) : plain;
      loss = loss + il0;
      float dl0 = 0.f;
      const bool dep0 = a.has_depth && a.coin[3];
      if (dep0) { dl0 = (float)(tot[1][2] / N1c); loss = loss + dl0; }
      a.terms[4] = il0; a.terms[5] = dl0;
      a.stats[4] = a.coin[2] ? (float)(2.0 / (3.0 * N1c)) : 0.f;
      a.stats[5] = 0.f;
      a.stats[6] = dep0 ? (float)(2.0 / N1c) / a.far : 0.f;
      a.stats[7] = 0.f;
      if (!a.coin[2]) { w1 += wall; w0 += wall; }
    }
    a.stats[0] = w1; a.stats[1] = w0;
    a.terms[0] = loss;
    a.terms[7] = 0.f;
    return;
  }
  float loss = 0.f;
  for (int lv = 0; lv < levels; ++lv) {
    const double s1 = tot[lv][0], s0 = tot[lv][1], sd = tot[lv][2];
    const double N1 = a.counts ? (double)a.counts[0] : tot[lv][3];
    const double N0 = a.counts ? (double)a.counts[1] : tot[lv][4];
    float il = (float)(s1 / (3.0 * N1));
    if (N0 > 0) il += a.coef * (float)(s0 / (3.0 * N0));
    const float dl = a.has_depth ? (float)(sd / N1) : 0.f;
    float pl = 0.f;
    for (int k = 0; k < a.P; ++k) pl += pshare[lv][k];
    loss += a.rgb_w * il;
    if (a.P > 0) loss += a.patch_w * pl;
    if (a.has_depth) loss = loss + a.depth_w * dl;
    a.terms[1 + 3 * lv] = il; a.terms[2 + 3 * lv] = dl; a.terms[3 + 3 * lv] = pl;
    a.stats[4 * lv + 0] = (float)(2.0 / (3.0 * N1));
    a.stats[4 * lv + 1] = N0 > 0 ? a.coef * (float)(2.0 / (3.0 * N0)) : 0.f;
    a.stats[4 * lv + 2] = (float)(2.0 / N1) / a.far;
    a.stats[4 * lv + 3] = 0.f;
  }
  if (levels == 1) { a.terms[4] = a.terms[5] = a.terms[6] = 0.f; }
  a.terms[0] = loss;
  a.terms[7] = 0.f;
}

}  // namespace

// img2mse (H:9): loss = mean((x - y)^2) over n elements + the gradient seed 2 (x - y) / n, one launch (replaces ATen's
// sub, pow, mean and their three backward kernels).  One workgroup, fixed order, fp64 accumulation.
namespace {
__global__ __launch_bounds__(T) void mse_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                           float* __restrict__ loss, float* __restrict__ d_x) {
  __shared__ double sh[T / 64];
  double s = 0;
  const float w = (float)(2.0 / (double)n);
  for (int64_t i = threadIdx.x; i < n; i += T) {
    const float d = x[i] - y[i];
    s += (double)(d * d);
    if (d_x) d_x[i] = w * d;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[0] = (float)(s / (double)n);
}
}  // namespace

extern "C" int cnerf_mse(const float* x, const float* y, int64_t n, float* loss, float* d_x, void* stream) {
  if (!x || !y || !loss || n <= 0) return CNERF_E_ARG;
  hipLaunchKernelGGL(mse_k, dim3(1), dim3(T), 0, cn_stream(stream), x, y, n, loss, d_x);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// img2mse_softLpmask (V:58; the `--softLpmask` branch of the loss, V:1663-1664 / V:1760-1761): every squared residual weighted by
// w = |d|^coef + 1, normalised by the DETACHED sum of the weights:  loss = sum(w d^2) / sum(w).  Gradient w.r.t. x (the denominator
// carries none): (dw/dd d^2 + 2 w d) / sum(w) with dw/dd = coef |d|^(coef - 1) sign(d)  ->  d (coef |d|^coef + 2 w) / sum(w).
// One workgroup, two sweeps (sums in fp64, fixed order), value + gradient seed in one launch like mse_k.
namespace {
__global__ __launch_bounds__(T) void soft_lp_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n, float coef,
                                               float* __restrict__ loss, float* __restrict__ d_x) {
  __shared__ double sh[T / 64];
  __shared__ double tot[2];
  double num = 0, den = 0;
  for (int64_t i = threadIdx.x; i < n; i += T) {
    const float d = x[i] - y[i];
    const float w = powf(fabsf(d), coef) + 1.f;
    num += (double)(w * (d * d));
    den += (double)w;
  }
  num = block_sum(num, sh);
  den = block_sum(den, sh);
  if (threadIdx.x == 0) { tot[0] = num; tot[1] = den; loss[0] = (float)(num / den); }
  __syncthreads();
  if (!d_x) return;
  const float inv = (float)(1.0 / tot[1]);
  for (int64_t i = threadIdx.x; i < n; i += T) {
    const float d = x[i] - y[i];
    const float p = powf(fabsf(d), coef);
    d_x[i] = (d * (coef * p + 2.f * (p + 1.f))) * inv;
  }
}
}  // namespace

extern "C" int cnerf_soft_lp_loss(const float* x, const float* y, int64_t n, float coef, float* loss, float* d_x, void* stream) {
  if (!x || !y || !loss || n <= 0 || !(coef > 0.f)) return CNERF_E_ARG;
  hipLaunchKernelGGL(soft_lp_k, dim3(1), dim3(T), 0, cn_stream(stream), x, y, n, coef, loss, d_x);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// Large inputs (whole images: img2mse of a rendered 756 x 1008 frame is 2.3 M elements): one workgroup per MSE_CHUNK elements writes
// its fp64 partial, a second single-workgroup stage sums the partials in index order — the value does not depend on the grid the
// first stage ran on or on the order its workgroups finished in.
namespace {
constexpr int64_t MSE_CHUNK = 16384;
__global__ __launch_bounds__(T) void mse_part_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                double* __restrict__ part, float* __restrict__ d_x) {
  __shared__ double sh[T / 64];
  const int64_t lo = (int64_t)blockIdx.x * MSE_CHUNK, hi = lo + MSE_CHUNK < n ? lo + MSE_CHUNK : n;
  const float w = (float)(2.0 / (double)n);
  double s = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += T) {
    const float d = x[i] - y[i];
    s += (double)(d * d);
    if (d_x) d_x[i] = w * d;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(T) void mse_fin_k(const double* __restrict__ part, int64_t nparts, int64_t n, float* __restrict__ loss) {
  __shared__ double sh[T / 64];
  double s = 0;
  for (int64_t i = threadIdx.x; i < nparts; i += T) s += part[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[0] = (float)(s / (double)n);
}
}  // namespace

extern "C" int64_t cnerf_mse_ws_floats(int64_t n) { return n <= 0 ? 0 : 2 * ((n + MSE_CHUNK - 1) / MSE_CHUNK); }

extern "C" int cnerf_mse_ws(const float* x, const float* y, int64_t n, float* loss, float* d_x, float* workspace, void* stream) {
  if (!x || !y || !loss || n <= 0) return CNERF_E_ARG;
  if (!workspace || n <= MSE_CHUNK) return cnerf_mse(x, y, n, loss, d_x, stream);
  if (((uintptr_t)workspace & 7) != 0) return CNERF_E_ARG;      // fp64 partials
  const int64_t nparts = (n + MSE_CHUNK - 1) / MSE_CHUNK;
  hipLaunchKernelGGL(mse_part_k, dim3((unsigned)nparts), dim3(T), 0, cn_stream(stream), x, y, n, reinterpret_cast<double*>(workspace),
                     d_x);
  CN_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_fin_k, dim3(1), dim3(T), 0, cn_stream(stream), reinterpret_cast<const double*>(workspace), nparts, n, loss);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// Large batches (whole images through the masked losses: evaluation, or a caller that does not shard): stage 1 — one workgroup per
// ML_CHUNK rays leaves its five fp64 partials; stage 2 — the same grid: every workgroup sums ALL partials in index order (a few
// hundred doubles: cheaper than a third launch), workgroup 0 writes the loss, each writes the gradient seeds of its own chunk.
// Fixed order at both stages: the value does not depend on scheduling.
namespace {
constexpr int64_t ML_CHUNK = 16384;
constexpr int64_t ML_MAX_PARTS = 1024;
__global__ __launch_bounds__(T) void masked_part_k(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                   const float* __restrict__ depth, const float* __restrict__ prior,
                                                   const float* __restrict__ mask, int64_t B, float far, double* __restrict__ part) {
  __shared__ double sh[T / 64];
  const int64_t lo = (int64_t)blockIdx.x * ML_CHUNK, hi = lo + ML_CHUNK < B ? lo + ML_CHUNK : B;
  double n1 = 0, n0 = 0, s1 = 0, s0 = 0, sd = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const bool in1 = m == 1.f, in0 = m == 0.f;
    float e = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = rgb[3 * i + c] - tgt[3 * i + c];
      e += d * d;
    }
    if (in1) { n1 += 1.0; s1 += (double)e; }
    if (in0) { n0 += 1.0; s0 += (double)e; }
    if (depth && in1) {
      const float d = depth[i] / far - prior[i] / far;
      sd += (double)(d * d);
    }
  }
  n1 = block_sum(n1, sh); n0 = block_sum(n0, sh);
  s1 = block_sum(s1, sh); s0 = block_sum(s0, sh); sd = block_sum(sd, sh);
  if (threadIdx.x == 0) {
    double* o = part + 5 * (int64_t)blockIdx.x;
    o[0] = s1; o[1] = s0; o[2] = sd; o[3] = n1; o[4] = n0;
  }
}
__global__ __launch_bounds__(T) void masked_fin_k(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                  const float* __restrict__ depth, const float* __restrict__ prior,
                                                  const float* __restrict__ mask, int64_t B, float far, float coef,
                                                  const float* __restrict__ counts, float g_scale, const double* __restrict__ part,
                                                  float* __restrict__ loss, float* __restrict__ d_rgb, float* __restrict__ d_depth) {
  double t[5] = {0, 0, 0, 0, 0};
  for (unsigned p = 0; p < gridDim.x; ++p)
    for (int k = 0; k < 5; ++k) t[k] += part[5 * (int64_t)p + k];
  const double N1 = counts ? (double)counts[0] : t[3];
  const double N0 = counts ? (double)counts[1] : t[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float l = (float)(t[0] / (3.0 * N1));
    if (N0 > 0) l += coef * (float)(t[1] / (3.0 * N0));
    loss[0] = l;
    loss[1] = depth ? (float)(t[2] / N1) : 0.f;
  }
  const float w1 = g_scale * (float)(2.0 / (3.0 * N1));
  const float w0 = N0 > 0 ? g_scale * coef * (float)(2.0 / (3.0 * N0)) : 0.f;
  const float wd = g_scale * (float)(2.0 / N1) / far;
  const int64_t lo = (int64_t)blockIdx.x * ML_CHUNK, hi = lo + ML_CHUNK < B ? lo + ML_CHUNK : B;
  for (int64_t i = lo + threadIdx.x; i < hi; i += T) {
    const float m = mask ? mask[i] : 1.f;
    const float w = m == 1.f ? w1 : (m == 0.f ? w0 : 0.f);
    if (d_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d_rgb[3 * i + c] = w * (rgb[3 * i + c] - tgt[3 * i + c]);
    }
    if (d_depth) d_depth[i] = (depth && m == 1.f) ? wd * (depth[i] / far - prior[i] / far) : 0.f;
  }
}
}  // namespace

extern "C" int64_t cnerf_loss_ws_floats(void) { return 2 * 5 * ML_MAX_PARTS; }

extern "C" int cnerf_masked_loss(const float* rgb, const float* target, const float* depth, const float* prior,
                                 const float* mask, int64_t B, float far, float coef, const float* counts,
                                 float g_scale, float* loss, float* d_rgb, float* d_depth, float* workspace,
                                 void* stream) {
  if (!rgb || !target || !loss || B <= 0 || (depth && !prior) || !(far > 0.f) || ((uintptr_t)workspace & 7) != 0) return CNERF_E_ARG;
  const int64_t nparts = (B + ML_CHUNK - 1) / ML_CHUNK;
  if (workspace && nparts > 1 && nparts <= ML_MAX_PARTS) {
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(masked_part_k, dim3((unsigned)nparts), dim3(T), 0, cn_stream(stream), rgb, target, depth, prior, mask, B, far,
                       part);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(masked_fin_k, dim3((unsigned)nparts), dim3(T), 0, cn_stream(stream), rgb, target, depth, prior, mask, B, far,
                       coef, counts, g_scale, part, loss, d_rgb, d_depth);
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  }
  hipLaunchKernelGGL(masked_loss_k, dim3(1), dim3(T), 0, cn_stream(stream), rgb, target, depth, prior, mask, B, far,
                     coef, counts, g_scale, loss, d_rgb, d_depth);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_patch_depth_loss(const float* depth_pred, const float* mono, int P, int n, float g_scale,
                                      float* loss, float* d_depth, void* stream) {
  if (!depth_pred || !mono || !loss || P <= 0 || P > 16 || n <= 0) return CNERF_E_ARG;
  hipLaunchKernelGGL(patch_depth_loss_k, dim3(1), dim3(64 * P), 0, cn_stream(stream), depth_pred, mono, P, n, g_scale,
                     loss, d_depth);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

namespace {
int closs_finish_impl(const cnerf_closs_sum* t, const int32_t* ss_coins, float* terms, float* stats, float* patch_d, void* stream,
                      int64_t seg_row = 0, const float* counts3 = nullptr) {
  if (!t || !terms || !stats || !t->ws_last || t->B <= 0 || t->P < 0 || t->P > 8 || (t->P > 0 && (t->n <= 0 || !t->mono || !t->depth_last)) ||
      (t->P > 0 && t->ws_coarse && !t->depth_coarse) || (t->has_depth && !(t->far > 0.f)) || (int64_t)t->P * t->n > t->B ||
      ((uintptr_t)t->ws_last & 7) != 0 || ((uintptr_t)t->ws_coarse & 7) != 0)
    return CNERF_E_ARG;
  ClossTail a;
  a.part[0] = reinterpret_cast<const double*>(t->ws_last);
  a.part[1] = reinterpret_cast<const double*>(t->ws_coarse);
  a.nparts = (int)cn_div_up(t->B, (int64_t)CN_CLOSS_RAYS_PER_WG);
  a.counts = t->counts; a.coef = t->coef; a.far = t->has_depth ? t->far : 1.f; a.rgb_w = t->rgb_w; a.depth_w = t->depth_w;
  a.patch_w = t->patch_w; a.has_depth = t->has_depth;
  a.depth[0] = t->depth_last; a.depth[1] = t->depth_coarse; a.mono = t->mono; a.P = t->P; a.n = t->n;
  a.patch_d[0] = (patch_d && t->P > 0) ? patch_d : nullptr;
  a.patch_d[1] = (patch_d && t->P > 0 && t->ws_coarse) ? patch_d + (int64_t)t->P * t->n : nullptr;
  a.terms = terms; a.stats = stats;
  a.ss = ss_coins ? (seg_row > 0 ? 2 : 1) : 0;
  for (int k = 0; k < 4; ++k) a.coin[k] = ss_coins ? (ss_coins[k] != 0) : 0;
  a.nparts1 = (int)(seg_row / CN_CLOSS_RAYS_PER_WG);
  a.counts3 = counts3;
  hipLaunchKernelGGL(closs_tail_k, dim3(1), dim3(T), 0, cn_stream(stream), a);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
}  // namespace

extern "C" int cnerf_closs_finish(const cnerf_closs_sum* t, float* terms, float* stats, float* patch_d, void* stream) {
  return closs_finish_impl(t, nullptr, terms, stats, patch_d, stream);
}

extern "C" int cnerf_closs_finish_ss(const cnerf_closs_sum* t, const int32_t* coins4, float* terms, float* stats, void* stream) {
  // (no patch term and no sharded counts in this mode: the reference's block has neither)
  if (!coins4 || !t || t->P != 0 || t->counts) return CNERF_E_ARG;
  return closs_finish_impl(t, coins4, terms, stats, nullptr, stream);
}

extern "C" int cnerf_closs_finish_ss2(const cnerf_closs_sum* t, const int32_t* coins4, int64_t seg_row, const float* counts3,
                                      float* terms12, float* stats16, void* stream) {
  // the levels' compositing launches ran over ONE batch of t->B rays = [primary | warped + padding] cut at seg_row (a multiple of the
  // 8 rays of a compositing workgroup, so that every partial belongs to one segment)
  if (!coins4 || !t || t->P != 0 || t->counts || seg_row <= 0 || seg_row >= t->B || seg_row % CN_CLOSS_RAYS_PER_WG != 0) return CNERF_E_ARG;
  return closs_finish_impl(t, coins4, terms12, stats16, nullptr, stream, seg_row, counts3);
}
