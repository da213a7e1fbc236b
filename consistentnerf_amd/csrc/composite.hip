// Alpha compositing along rays (raw2outputs, reference R:265-308 / V:392-438) and its backward.
// One wave64 per ray: lane l owns the C = ceil(S/64) consecutive samples [l*C, l*C+C).  The exclusive
// transmittance product is a lane-local sequential product + one wave scan, carried in fp64 and
// rounded to fp32 per sample (the CPU reference's cumprod accumulates fp32 inputs in fp64).
// HBM-bound: 24 B per ray-sample forward (raw 16 + z 4 in, weights 4 out), 40 B backward.
#include "raygen.hpp"

namespace {

constexpr int WAVES = 4;

// img2mse(rgb_map, target) (H:9, R:769-775) folded into the compositing of a level.  Forward: every workgroup leaves the fp64 sum of
// its rays' squared colour errors in part[blockIdx.x]; the workgroup that takes the last ticket of `counter` sums the partials IN
// INDEX ORDER (the value does not depend on which workgroup finished last), writes loss[0] = fp32(sum / n) (+ loss_add[0]: the other
// level's term, `img_loss + img_loss0` in fp32 like R:775) and re-arms the counter.  Backward: the seed d loss / d rgb_map =
// (2 / n) (rgb_map - target) * g[0] is formed per ray in registers — same operations, same order as img2mse's own backward, so the
// fused and the separate paths agree bit for bit.
constexpr int MSE_WAVES = CN_CLOSS_RAYS_PER_WG;   // rays per workgroup of the loss form (common.hpp: loss.hip indexes the partials)
constexpr unsigned MSE_GROUP = 64;        // workgroups per first-level ticket counter
constexpr unsigned MSE_CTR_STRIDE = 64;   // uint32 words between counters (256 B: one counter per cache line / channel)
constexpr unsigned MSE_CTR_WORDS = 16384; // the caller's zeroed counter block: top counter + up to 255 group counters
struct MseFwd {
  const float* tgt;      // [B,3]; nullptr = no loss
  double* part;          // [gridDim.x]
  unsigned* counter;     // MSE_CTR_WORDS words, zero on entry, zero on exit
  float* loss;           // [1]
  const float* loss_add; // [1] or nullptr
  double n;              // elements of the mean (3 B)
};
struct MseBwd {
  const float* rgb;      // forward rgb_map [B,3]; nullptr = seeds come from g_rgb
  const float* tgt;
  const float* g;        // upstream gradient of the loss (device scalar) or nullptr = 1
  float w;               // fp32(2 / n)
};

// ConsistentNeRF's masked losses (V:1645-1648, V:1737, V:1786-1788, V:1865) folded into the compositing of a level the same way —
// without tickets: every workgroup leaves FIVE fp64 partials (squared colour error over the rays with m == 1 and with m == 0, squared
// depth error (depth / far - prior / far)^2 over m == 1, and the two counts) in part[k * gridDim.x + blockIdx.x]; the step's loss
// tail (loss.hip closs_tail_k, the next launch on the stream: it also evaluates the monocular patch term, which needs whole
// depth maps) sums them in index order, normalises by the (possibly global) counts and leaves the three seed weights the backward
// uses.  Backward: seed_rgb = (w_m (rgb - target)) * g_rgb, seed_depth = (w_d (depth / far - prior / far)) * g_depth (+ the patch
// term's d_depth * g_patch on the patch rays) — the operations of masked_loss_k / patch_depth_loss_k and autograd's `d * g`.
struct ClossFwd {
  const float* tgt;      // [B,3]
  const float* mask;     // [B] or nullptr (every ray in the m == 1 set)
  const float* prior;    // [B] or nullptr (no depth term)
  float far;
  double* part;          // [5][gridDim.x]
};
struct ClossBwd {
  const float* rgb;      // forward rgb_map [B,3]
  const float* depth;    // forward depth_map [B] (with prior)
  const float* tgt;
  const float* mask;
  const float* prior;
  const float* stats;    // device [3]: w1, w0, wd (closs_tail_k); with seg_row > 0: [2][4], the second set for rays >= seg_row
  int64_t seg_row;       // 0, or the first ray of the batch's second segment (cnerf_closs::seg_row: the one-render a15 step)
  const float* g;        // upstream gradient of the total loss (device scalar) or nullptr = 1
  const float* patch_d;  // [n_patch] d patch_loss / d depth of this level, or nullptr
  int64_t n_patch;
  float far, rgb_w, depth_w, patch_w;
};

struct Sample {
  float e;      // exp(-relu(sigma)*dist)
  float alpha;  // 1 - e
  float x;      // 1 - alpha + 1e-10   (factor of the transmittance product)
  float r, g, b;  // sigmoid colours
  float z, dist, sig;
  bool live;
};

template <int C>
__device__ __forceinline__ void load_samples(Sample (&sm)[C], const float* __restrict__ raw, int ch,
                                             const float* __restrict__ zrow, const float* __restrict__ nrow,
                                             float dnorm, int S, int lane) {
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const int s = lane * C + j;
    Sample& q = sm[j];
    q.live = s < S;
    if (!q.live) {
      q.e = 1.f; q.alpha = 0.f; q.x = 1.f; q.r = q.g = q.b = 0.f; q.z = 0.f; q.dist = 0.f; q.sig = 0.f;
      continue;
    }
    float c0, c1, c2, sg;
    if (ch == 4) {
      const float4 v = *reinterpret_cast<const float4*>(raw + (int64_t)s * 4);
      c0 = v.x; c1 = v.y; c2 = v.z; sg = v.w;
    } else {
      const float* p = raw + (int64_t)s * ch;
      c0 = p[0]; c1 = p[1]; c2 = p[2]; sg = p[3];
    }
    q.z = zrow[s];
    float d = (s + 1 < S) ? (zrow[s + 1] - q.z) : 1e10f;   // R:280-281
    q.dist = d * dnorm;                                      // R:283
    if (nrow) sg = sg + nrow[s];                             // R:296
    q.sig = sg;
    const float act = sg > 0.f ? sg : 0.f;
    q.e = expf(-act * q.dist);
    q.alpha = 1.f - q.e;
    q.x = 1.f - q.alpha + 1e-10f;                            // R:298
    q.r = 1.f / (1.f + expf(-c0));
    q.g = 1.f / (1.f + expf(-c1));
    q.b = 1.f / (1.f + expf(-c2));
  }
}

// exclusive product scan over all samples of the ray; T[j] = fp32(prod_{k<s} x_k)
template <int C>
__device__ __forceinline__ void transmittance(const Sample (&sm)[C], float (&T)[C], int lane) {
  double p = 1.0;
#pragma unroll
  for (int j = 0; j < C; ++j) p *= (double)sm[j].x;
  double incl = p;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double v = __shfl_up(incl, o, 64);
    if (lane >= o) incl *= v;
  }
  double run = __shfl_up(incl, 1, 64);
  if (lane == 0) run = 1.0;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    T[j] = (float)run;
    run *= (double)sm[j].x;
  }
}

__device__ __forceinline__ float ray_norm(const float* __restrict__ ray) {
  const float dx = ray[3], dy = ray[4], dz = ray[5];
  return sqrtf(dx * dx + dy * dy + dz * dz);
}

// one ray by one wave64; returns (lane 0) the squared colour error against tgt (0 without a target)
template <int C>
__device__ __forceinline__ double composite_ray(const float* __restrict__ raw, int ch, const float* __restrict__ z,
                                               const float* __restrict__ rays, int rs, const float* __restrict__ noise, int64_t b,
                                               int S, int white, float* __restrict__ rgb, float* __restrict__ disp,
                                               float* __restrict__ acc, float* __restrict__ depth, float* __restrict__ weights,
                                               const RayGenDev& cam, const float* __restrict__ tgt, float (&out4)[4]) {
  const int lane = threadIdx.x & 63;
  double err = 0.0;
  Sample sm[C];
  float T[C];
  float dn;
  if (rays != nullptr) {
    dn = ray_norm(rays + b * rs);
  } else {   // the ray of a camera, generated here (raygen.hpp): R:280-283 scales the sample distances by |rays_d|
    float o[3], d[3], v[3];
    cn_gen_ray(cam, cam.first + b, o, d, v);
    dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  }
  load_samples<C>(sm, raw + b * S * ch, ch, z + b * S, noise ? noise + b * S : nullptr, dn, S, lane);
  transmittance<C>(sm, T, lane);
  double sr = 0, sg = 0, sb = 0, sd = 0, sa = 0;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const float w = sm[j].alpha * T[j];
    if (sm[j].live) {
      if (weights) weights[b * S + lane * C + j] = w;
      sr += (double)(w * sm[j].r);
      sg += (double)(w * sm[j].g);
      sb += (double)(w * sm[j].b);
      sd += (double)(w * sm[j].z);
      sa += (double)w;
    }
  }
  sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
  if (lane == 0) {
    const float fa = (float)sa, fd = (float)sd;
    float r = (float)sr, g = (float)sg, bl = (float)sb;
    if (white) { const float bg = 1.f - fa; r += bg; g += bg; bl += bg; }   // R:305-306
    out4[0] = r; out4[1] = g; out4[2] = bl; out4[3] = fd;
    if (rgb) { rgb[b * 3 + 0] = r; rgb[b * 3 + 1] = g; rgb[b * 3 + 2] = bl; }
    if (acc) acc[b] = fa;
    if (depth) depth[b] = fd;
    if (disp) {
      const float q = fd / fa;                          // 0/0 -> NaN propagates like torch.max (R:302)
      disp[b] = (q != q) ? q : 1.f / fmaxf(1e-10f, q);
    }
    if (tgt) {                                          // (x - y)^2 per element in fp32, summed in fp64 like mse_k (loss.hip)
      const float d0 = r - tgt[b * 3 + 0], d1 = g - tgt[b * 3 + 1], d2 = bl - tgt[b * 3 + 2];
      err = (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2);
    }
  }
  return err;
}

// WV = rays (waves) per workgroup: WAVES for the plain form; MSE_WAVES for the loss form (fewer, fatter partials and tickets)
template <int C, int WV>
__global__ __launch_bounds__(WV * 64) void composite_fwd_k(const float* __restrict__ raw, int ch,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ rays, int rs,
                                                              const float* __restrict__ noise, int64_t B, int S,
                                                              int white, float* __restrict__ rgb,
                                                              float* __restrict__ disp, float* __restrict__ acc,
                                                              float* __restrict__ depth, float* __restrict__ weights,
                                                              RayGenDev cam, MseFwd mse, ClossFwd cl) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WV + (threadIdx.x >> 6);
  float o4[4] = {0.f, 0.f, 0.f, 0.f};
  if (cl.tgt != nullptr) {
    // masked-loss variant: per-ray terms in lane 0 (the arithmetic of masked_loss_k: e = d0^2 + d1^2 + d2^2 in fp32, sums in
    // fp64), five partials per workgroup summed over its waves in wave order; no tickets (see ClossFwd)
    __shared__ double sq[WV][5];
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (b < B) {
      composite_ray<C>(raw, ch, z, rays, rs, noise, b, S, white, rgb, disp, acc, depth, weights, cam, nullptr, o4);
      if (lane == 0) {
        const float m = cl.mask ? cl.mask[b] : 1.f;
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = o4[c] - cl.tgt[b * 3 + c];
          e += d * d;
        }
        if (m == 1.f) { t[0] = (double)e; t[3] = 1.0; }
        if (m == 0.f) { t[1] = (double)e; t[4] = 1.0; }
        if (cl.prior && m == 1.f) {
          const float d = o4[3] / cl.far - cl.prior[b] / cl.far;
          t[2] = (double)(d * d);
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sq[threadIdx.x >> 6][k] = t[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
      double s = 0.0;
      for (int w = 0; w < WV; ++w) s += sq[w][threadIdx.x];
      cl.part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
    return;
  }
  if (mse.tgt != nullptr) {
    // loss variant: every wave reaches the workgroup reduction below (a wave past the last ray contributes 0); after it only wave 0
    // stays for the tickets (the others retire: a workgroup that waits ~2 us for two memory round trips must not hold 16 wave slots)
    __shared__ double sq[WV];
    double e = 0.0;
    if (b < B) e = composite_ray<C>(raw, ch, z, rays, rs, noise, b, S, white, rgb, disp, acc, depth, weights, cam, mse.tgt, o4);
    if (lane == 0) sq[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int is_last = 0;
    if (lane == 0) {
      double s = 0.0;
      for (int w = 0; w < WV; ++w) s += sq[w];
      // Publish the partial, then take a ticket.  Everything that crosses workgroups here is an agent-scope ATOMIC access: the
      // partial is stored write-through (`global_store_dwordx2 ... sc1`), the reducer loads it with sc1, so neither side needs an
      // L2 write-back / invalidate (the agent-scope release fence = `buffer_wbl2 sc1` made every workgroup write back its XCD's
      // L2: 44 us per launch).  What the ticket DOES need is the store's completion: vmcnt(0) between the store and the atomic.
      // A workgroup-scope release fence compiles to NOTHING on gfx950 (the ticket could overtake the partial), so the
      // wait is written out; `scripts/isa_ticket_check.py` asserts the s_waitcnt vmcnt(0) sits between the two in the ISA.
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mse.part) + blockIdx.x, (unsigned long long)__double_as_longlong(s),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(CN_MSE_HEAVY_FENCE)
      __threadfence();
#elif !defined(CN_MSE_RELEASE_TICKET)
      // vmcnt(0): the sc1 store has been acknowledged by the memory side.  Written as inline assembly with a "memory" clobber
      // (ADVICE r05): the mnemonic carries no architecture-specific immediate (the builtin's 0x0f70 is the gfx9 encoding), and the
      // clobber is a compiler-level barrier on BOTH sides — the partial's store cannot sink below the wait, the ticket RMW cannot be
      // hoisted above it — which the s_waitcnt builtin (IntrNoMem) + a sched_barrier after it did not state.  The order in the ISA
      // is asserted at BUILD time (consistentnerf_amd/build.py -> scripts/isa_ticket_check.py), not only by a test.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      // Two-level ticket: same-address atomics at agent scope serialise at ~40 ns each (measured: 8192 workgroups on ONE counter
      // cost 0.36 ms), so a workgroup first tickets inside its group of MSE_GROUP (one counter per group, 256 B apart); the last of
      // a group tickets at the top.  <= MSE_GROUP + ngroups serialised atomics instead of gridDim.x.
      const unsigned grp = blockIdx.x / MSE_GROUP, ngroups = (gridDim.x + MSE_GROUP - 1) / MSE_GROUP;
      const unsigned in_grp = grp + 1 == ngroups ? gridDim.x - grp * MSE_GROUP : MSE_GROUP;
      unsigned* gctr = mse.counter + MSE_CTR_STRIDE * (1 + grp);
      bool last = false;
#ifdef CN_MSE_RELEASE_TICKET   // ablation: the textbook form (release-ordered agent-scope RMW; measured in profiles/r05_ticket_*.txt)
      constexpr int TICKET_ORDER = __ATOMIC_ACQ_REL;
#else
      constexpr int TICKET_ORDER = __ATOMIC_RELAXED;
#endif
      if (__hip_atomic_fetch_add(gctr, 1u, TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1) {
        __hip_atomic_store(gctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-arm (nobody else touches it any more)
        last = __hip_atomic_fetch_add(mse.counter, 1u, TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
      }
      is_last = last;
    }
    is_last = __builtin_amdgcn_readfirstlane(is_last);
    if (is_last) {
      // The tickets form a chain of agent-scope RMWs on device memory: the last ticket was granted after every other workgroup's
      // (its partial complete before it, above).  The partial loads below are agent-scope atomic loads (sc1: served by the memory
      // side, never by this XCD's L2 or the vector L1) and depend on is_last, which came back from the ticket atomic, so they cannot
      // be issued before it returned.
#ifdef CN_MSE_HEAVY_FENCE
      __threadfence();
#endif
      double s = 0.0;
      for (unsigned i = lane; i < gridDim.x; i += 64)
        s += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(mse.part) + i, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
      s = wave_sum(s);
      if (lane == 0) {
        float l = (float)(s / mse.n);
        if (mse.loss_add) l = l + mse.loss_add[0];
        mse.loss[0] = l;
        __hip_atomic_store(mse.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  if (b >= B) return;
  composite_ray<C>(raw, ch, z, rays, rs, noise, b, S, white, rgb, disp, acc, depth, weights, cam, nullptr, o4);
}

template <int C>
__global__ __launch_bounds__(WAVES * 64) void composite_bwd_k(const float* __restrict__ raw, int ch,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ rays, int rs,
                                                              const float* __restrict__ noise, int64_t B, int S,
                                                              int white, const float* __restrict__ g_rgb,
                                                              const float* __restrict__ g_disp,
                                                              const float* __restrict__ g_acc,
                                                              const float* __restrict__ g_depth,
                                                              float* __restrict__ d_raw, MseBwd mse, ClossBwd cl) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  Sample sm[C];
  float T[C];
  load_samples<C>(sm, raw + b * S * ch, ch, z + b * S, noise ? noise + b * S : nullptr, ray_norm(rays + b * rs), S,
                  lane);
  transmittance<C>(sm, T, lane);
  float gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f;
  if (g_rgb) { gr = g_rgb[b * 3 + 0]; gg = g_rgb[b * 3 + 1]; gb = g_rgb[b * 3 + 2]; }
  if (mse.rgb) {   // the img2mse seed of this level, formed here: (w * (x - y)) * g — the operations of mse_k then `d_x * g`
    const float g0 = mse.g ? mse.g[0] : 1.f;
    gr += (mse.w * (mse.rgb[b * 3 + 0] - mse.tgt[b * 3 + 0])) * g0;
    gg += (mse.w * (mse.rgb[b * 3 + 1] - mse.tgt[b * 3 + 1])) * g0;
    gb += (mse.w * (mse.rgb[b * 3 + 2] - mse.tgt[b * 3 + 2])) * g0;
  }
  if (g_depth) gd = g_depth[b];
  if (cl.rgb) {    // the masked rgb / depth seeds (+ the patch term's) of this level, formed here (see ClossBwd)
    const float g0 = cl.g ? cl.g[0] : 1.f;
    const float m = cl.mask ? cl.mask[b] : 1.f;
    const float* const st = cl.stats + ((cl.seg_row > 0 && b >= cl.seg_row) ? 4 : 0);
    const float w = m == 1.f ? st[0] : (m == 0.f ? st[1] : 0.f);
    const float g_rgb_l = cl.rgb_w * g0;
    gr += (w * (cl.rgb[b * 3 + 0] - cl.tgt[b * 3 + 0])) * g_rgb_l;
    gg += (w * (cl.rgb[b * 3 + 1] - cl.tgt[b * 3 + 1])) * g_rgb_l;
    gb += (w * (cl.rgb[b * 3 + 2] - cl.tgt[b * 3 + 2])) * g_rgb_l;
    if (cl.prior) {
      const float dd = m == 1.f ? st[2] * (cl.depth[b] / cl.far - cl.prior[b] / cl.far) : 0.f;
      gd += dd * (cl.depth_w * g0);
    }
    if (cl.patch_d && b < cl.n_patch) gd += cl.patch_d[b] * (cl.patch_w * g0);
  }
  if (g_acc) ga = g_acc[b];
  if (white) ga -= (gr + gg + gb);
  if (g_disp) {
    // disp = 1/max(1e-10, depth/acc): needs the forward totals
    double sd = 0, sa = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float w = sm[j].alpha * T[j];
      sd += (double)(w * sm[j].z);
      sa += (double)w;
    }
    const float fd = (float)wave_sum(sd), fa = (float)wave_sum(sa);
    const float q = fd / fa;
    if (q > 1e-10f) {
      const float gq = -g_disp[b] / (q * q);
      gd += gq / fa;
      ga += -gq * fd / (fa * fa);
    }
  }
  // v_i = dL/dw_i ; suffix sums S_i = sum_{k>i} w_k v_k (reverse exclusive scan, fp64)
  float v[C];
  double tot = 0.0;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    v[j] = gr * sm[j].r + gg * sm[j].g + gb * sm[j].b + gd * sm[j].z + ga;
    tot += (double)(sm[j].alpha * T[j]) * (double)v[j];
  }
  double incl = tot;   // inclusive suffix over lanes >= l
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_down(incl, o, 64);
    if (lane + o < 64) incl += t;
  }
  double suf = incl - tot;   // sum over lanes > l
#pragma unroll
  for (int j = C - 1; j >= 0; --j) {
    const float w = sm[j].alpha * T[j];
    // cumprod backward (as autograd: reverse-cumsum(grad*out)/input) -> dL/dalpha
    const float dalpha = T[j] * v[j] - (float)(suf / (double)sm[j].x);
    suf += (double)w * (double)v[j];
    if (!sm[j].live) continue;
    const float dsig = sm[j].sig > 0.f ? dalpha * sm[j].dist * sm[j].e : 0.f;
    const float dr = gr * w * sm[j].r * (1.f - sm[j].r);
    const float dg = gg * w * sm[j].g * (1.f - sm[j].g);
    const float db = gb * w * sm[j].b * (1.f - sm[j].b);
    float* o = d_raw + (b * S + lane * C + j) * ch;
    if (ch == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(dr, dg, db, dsig);
    } else {
      o[0] = dr; o[1] = dg; o[2] = db; o[3] = dsig;
      for (int c = 4; c < ch; ++c) o[c] = 0.f;
    }
  }
}

template <typename F>
int dispatch_c(int S, F f) {
  const int C = (S + 63) / 64;
  if (C <= 1) return f(std::integral_constant<int, 1>());
  if (C == 2) return f(std::integral_constant<int, 2>());
  if (C == 3) return f(std::integral_constant<int, 3>());
  if (C == 4) return f(std::integral_constant<int, 4>());
  if (C <= 8) return f(std::integral_constant<int, 8>());
  if (C <= 16) return f(std::integral_constant<int, 16>());
  return CNERF_E_UNSUPPORTED;
}

}  // namespace

extern "C" int cnerf_composite_fwd(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                   const float* noise, int64_t B, int S, int white_bkgd, float* rgb, float* disp,
                                   float* acc, float* depth, float* weights, void* stream) {
  if (!raw || !z || !rays || B < 0 || S <= 0 || raw_ch < 4 || ray_stride < 6) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  return dispatch_c(S, [&](auto c) -> int {
    constexpr int C = decltype(c)::value;
    hipLaunchKernelGGL((composite_fwd_k<C, WAVES>), dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0,
                       cn_stream(stream), raw, raw_ch, z, rays, ray_stride, noise, B, S, white_bkgd, rgb, disp, acc,
                       depth, weights, cn_no_raygen(), MseFwd{}, ClossFwd{});
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

// compositing for the rays of a camera generated in-kernel; used by cnerf_render_fwd_cam
int cn_composite_fwd_cam(const float* raw, int raw_ch, const float* z, const RayGenDev& cam, const float* noise, int64_t B,
                         int S, int white_bkgd, float* rgb, float* disp, float* acc, float* depth, float* weights,
                         hipStream_t st) {
  if (!raw || !z || B < 0 || S <= 0 || raw_ch < 4) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  return dispatch_c(S, [&](auto c) -> int {
    constexpr int C = decltype(c)::value;
    hipLaunchKernelGGL((composite_fwd_k<C, WAVES>), dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, st, raw, raw_ch, z,
                       (const float*)nullptr, 0, noise, B, S, white_bkgd, rgb, disp, acc, depth, weights, cam, MseFwd{}, ClossFwd{});
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

extern "C" int cnerf_composite_bwd(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                   const float* noise, int64_t B, int S, int white_bkgd, const float* g_rgb,
                                   const float* g_disp, const float* g_acc, const float* g_depth, float* d_raw,
                                   void* stream) {
  if (!raw || !z || !rays || !d_raw || B < 0 || S <= 0 || raw_ch < 4 || ray_stride < 6) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  return dispatch_c(S, [&](auto c) -> int {
    constexpr int C = decltype(c)::value;
    hipLaunchKernelGGL((composite_bwd_k<C>), dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0,
                       cn_stream(stream), raw, raw_ch, z, rays, ray_stride, noise, B, S, white_bkgd, g_rgb, g_disp,
                       g_acc, g_depth, d_raw, MseBwd{}, ClossBwd{});
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

// ---- compositing with img2mse(rgb_map, target) folded in (R:769-775): see MseFwd / MseBwd above ---------------------------------
extern "C" int64_t cnerf_composite_mse_ws_floats(int64_t B) { return B <= 0 ? 0 : 2 * cn_div_up(B, WAVES) + 2; }
extern "C" int64_t cnerf_composite_mse_counter_words(void) { return MSE_CTR_WORDS; }
extern "C" int64_t cnerf_composite_mse_max_rays(void) { return (int64_t)(MSE_CTR_WORDS / MSE_CTR_STRIDE - 1) * MSE_GROUP * MSE_WAVES; }

extern "C" int cnerf_composite_fwd_mse(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                       const float* noise, int64_t B, int S, int white_bkgd, const float* target,
                                       const float* loss_add, float* rgb, float* disp, float* acc, float* depth, float* weights,
                                       float* loss, float* workspace, unsigned* counter, void* stream) {
  if (!raw || !z || !rays || !target || !rgb || !loss || !workspace || !counter || B <= 0 || S <= 0 || raw_ch < 4 ||
      ray_stride < 6 || ((uintptr_t)workspace & 7) != 0)
    return CNERF_E_ARG;
  if (B > cnerf_composite_mse_max_rays()) return CNERF_E_UNSUPPORTED;
  MseFwd m;
  m.tgt = target; m.part = reinterpret_cast<double*>(workspace); m.counter = counter; m.loss = loss; m.loss_add = loss_add;
  m.n = 3.0 * (double)B;
  return dispatch_c(S, [&](auto c) -> int {
    constexpr int C = decltype(c)::value;
    hipLaunchKernelGGL((composite_fwd_k<C, MSE_WAVES>), dim3((unsigned)cn_div_up(B, MSE_WAVES)), dim3(MSE_WAVES * 64), 0,
                       cn_stream(stream), raw, raw_ch, z, rays, ray_stride, noise, B, S, white_bkgd, rgb, disp, acc, depth, weights,
                       cn_no_raygen(), m, ClossFwd{});
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

extern "C" int cnerf_composite_bwd_mse(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                       const float* noise, int64_t B, int S, int white_bkgd, const float* rgb, const float* target,
                                       const float* g_loss, float* d_raw, void* stream) {
  if (!raw || !z || !rays || !rgb || !target || !d_raw || B <= 0 || S <= 0 || raw_ch < 4 || ray_stride < 6) return CNERF_E_ARG;
  MseBwd m;
  m.rgb = rgb; m.tgt = target; m.g = g_loss; m.w = (float)(2.0 / (3.0 * (double)B));
  return dispatch_c(S, [&](auto c) -> int {
    constexpr int C = decltype(c)::value;
    hipLaunchKernelGGL((composite_bwd_k<C>), dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, cn_stream(stream), raw, raw_ch,
                       z, rays, ray_stride, noise, B, S, white_bkgd, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, d_raw, m, ClossBwd{});
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

// ---- compositing with ConsistentNeRF's masked rgb / depth losses folded in (V:1645-1865): see ClossFwd / ClossBwd above ----------
extern "C" int64_t cnerf_closs_ws_floats(int64_t B) { return B <= 0 ? 0 : 10 * cn_div_up(B, MSE_WAVES); }

extern "C" int cnerf_composite_fwd_closs(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                         const float* noise, int64_t B, int S, int white_bkgd, const cnerf_closs* L, float* rgb,
                                         float* disp, float* acc, float* depth, float* weights, float* workspace, void* stream) {
  if (!raw || !z || !rays || !L || !L->target || !rgb || !workspace || B <= 0 || S <= 0 || raw_ch < 4 || ray_stride < 6 ||
      ((uintptr_t)workspace & 7) != 0 || (L->prior && (!depth || !(L->far > 0.f))))
    return CNERF_E_ARG;
  ClossFwd c;
  c.tgt = L->target; c.mask = L->mask; c.prior = L->prior; c.far = L->far; c.part = reinterpret_cast<double*>(workspace);
  return dispatch_c(S, [&](auto cc) -> int {
    constexpr int C = decltype(cc)::value;
    hipLaunchKernelGGL((composite_fwd_k<C, MSE_WAVES>), dim3((unsigned)cn_div_up(B, MSE_WAVES)), dim3(MSE_WAVES * 64), 0,
                       cn_stream(stream), raw, raw_ch, z, rays, ray_stride, noise, B, S, white_bkgd, rgb, disp, acc, depth, weights,
                       cn_no_raygen(), MseFwd{}, c);
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}

extern "C" int cnerf_composite_bwd_closs(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                                         const float* noise, int64_t B, int S, int white_bkgd, const cnerf_closs* L, const float* rgb,
                                         const float* depth, const float* stats, const float* g_loss, float rgb_w, float depth_w,
                                         float patch_w, const float* patch_d, int64_t n_patch_rays, float* d_raw, void* stream) {
  if (!raw || !z || !rays || !L || !L->target || !rgb || !stats || !d_raw || B <= 0 || S <= 0 || raw_ch < 4 || ray_stride < 6 ||
      (L->prior && (!depth || !(L->far > 0.f))) || n_patch_rays < 0 || n_patch_rays > B)
    return CNERF_E_ARG;
  ClossBwd c;
  c.rgb = rgb; c.depth = depth; c.tgt = L->target; c.mask = L->mask; c.prior = L->prior; c.stats = stats; c.g = g_loss;
  c.seg_row = L->seg_row > 0 && L->seg_row < B ? L->seg_row : 0;
  c.patch_d = n_patch_rays > 0 ? patch_d : nullptr; c.n_patch = n_patch_rays; c.far = L->far; c.rgb_w = rgb_w; c.depth_w = depth_w;
  c.patch_w = patch_w;
  return dispatch_c(S, [&](auto cc) -> int {
    constexpr int C = decltype(cc)::value;
    hipLaunchKernelGGL((composite_bwd_k<C>), dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, cn_stream(stream), raw, raw_ch,
                       z, rays, ray_stride, noise, B, S, white_bkgd, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, d_raw, MseBwd{}, c);
    CN_CHECK_LAUNCH();
    return CNERF_OK;
  });
}
