// Sample placement along rays: stratified coarse depths, inverse-CDF resampling, sorted merge.
// Replaces render_rays R:355-382 / R:395-399,415 and sample_pdf H:206-250 of the reference.
// One wave64 per ray for the scan/search/sort parts; everything a ray needs lives in LDS/registers.
#include "common.hpp"
#include "rng.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// z[b,i] = near*(1-t_i) + far*t_i   (or the inverse-depth form), then the stratified jitter
// z = lower + (upper-lower)*t_rand with bin edges at the midpoints (R:360-382).  Operation order and
// roundings are the reference's (compiled with -ffp-contract=off: no FMA contraction).
__device__ __forceinline__ float zlin(float near, float far, float t, int lindisp) {
  if (!lindisp) return near * (1.f - t) + far * t;
  return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
}

__global__ void coarse_z_k(const float* __restrict__ rays, int rs, int64_t B, int Nc,
                           const float* __restrict__ t_vals, const float* __restrict__ t_rand, int lindisp,
                           float* __restrict__ z, float cam_near, float cam_far, int use_rng, CnRngK rngk) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Nc) return;
  int64_t b = idx / Nc;
  int i = (int)(idx - b * Nc);
  // (rays == nullptr: the rays of a camera, generated in the consumers — near / far are the camera's constants)
  float near = rays ? rays[b * rs + 6] : cam_near, far = rays ? rays[b * rs + 7] : cam_far;
  float zi = zlin(near, far, t_vals[i], lindisp);
  if (t_rand != nullptr || use_rng) {
    float lower, upper;
    if (i == 0) lower = zi;
    else lower = .5f * (zi + zlin(near, far, t_vals[i - 1], lindisp));
    if (i == Nc - 1) upper = zi;
    else upper = .5f * (zlin(near, far, t_vals[i + 1], lindisp) + zi);
    // (use_rng: the jitter is generated here — element (row0 + b, i) of the global [rays, Nc] stream, rng.hpp)
    const float t = use_rng ? CnRngDev(rngk).uniform(b, Nc, i) : t_rand[idx];
    zi = lower + (upper - lower) * t;
  }
  z[idx] = zi;
}

// ------------------------------------------------------------------------------------------------
constexpr int MAX_NB = 256;     // CDF entries per ray
constexpr int MAX_ALL = 1024;   // Nc + Nf
constexpr int WAVES = 4;

struct PdfLds {
  float cdf[WAVES][MAX_NB];
  float bins[WAVES][MAX_NB];
};

// torch.sum(x, -1) of one row of n fp32 values WITH THE ASSOCIATION OF ATen's CPU KERNEL (aten/src/ATen/native/cpu/
// SumKernel.cpp, fp32 accumulators): the contiguous inner reduction runs on 8-float vectors — `row_sum` keeps 4 vector
// accumulators (vector i of a group of four goes to accumulator i & 3, groups in order; leftover vectors to accumulator 0;
// then acc0 += acc1, acc2, acc3), the n % 8 tail is summed sequentially from zero, and the 8 lanes of the vector sum are added
// to it in lane order; rows shorter than one vector take the same `row_sum` on scalars.  (Rows of >= 512 elements would add
// ATen's cascade levels; MAX_NB rules them out.)  This is the pdf normaliser of sample_pdf (H:213): one fp32 ulp of it
// decides on which side of 1.0 the last CDF entry lands, i.e. the index of every sample at a CDF tie — the u = 1.0 sample of
// each ray on the deterministic test-time path.  Verified against torch.sum on the 8192 rows of the bulk fixture (0 differing
// bits) and, end to end, on its 2 x 524 288 indices (tests/golden/sample_pdf_bulk.npz).  Wave-uniform result.
template <typename AFn>
__device__ __forceinline__ float aten_row_sum(int n, int lane, AFn a) {
  const int V = n >= 8 ? 8 : 1;
  const int nv = n / V, size_ilp = nv >> 2;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  if (lane < V) {
    for (int i = 0; i < size_ilp; ++i) {
      p0 += a((4 * i + 0) * V + lane);
      p1 += a((4 * i + 1) * V + lane);
      p2 += a((4 * i + 2) * V + lane);
      p3 += a((4 * i + 3) * V + lane);
    }
    for (int i = 4 * size_ilp; i < nv; ++i) p0 += a(i * V + lane);
    p0 += p1;
    p0 += p2;
    p0 += p3;
  }
  float acc = 0.f;
  for (int k = nv * V; k < n; ++k) acc += a(k);
  for (int l = 0; l < V; ++l) acc += __shfl(p0, l, 64);
  return acc;
}

// Builds cdf[0..Nb-1] and bins[0..Nb-1] for one ray in LDS.  w(j), j<Nb-1, is the raw weight;
// pdf = (w+1e-5)/sum with the sum associated as torch.sum does on the CPU (aten_row_sum), cdf = [0, cumsum(pdf)] with the
// running sum carried in fp64 and each entry rounded to fp32 (what the CPU reference's cumsum produces).
template <typename WFn, typename BFn>
__device__ __forceinline__ void build_cdf(float* cdf, float* bins, int Nb, int lane, WFn w, BFn bin) {
  const int nw = Nb - 1;
  // per-lane contiguous chunk so the scan is lane-local + one wave scan
  const int C = (nw + 63) >> 6;
  const float sum = aten_row_sum(nw, lane, [&](int k) { return w(k) + 1e-5f; });
  double run = 0.0;
  for (int j = 0; j < C; ++j) {
    int k = lane * C + j;
    if (k < nw) run += (double)((w(k) + 1e-5f) / sum);
  }
  // inclusive wave scan of the lane totals
  double incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  double acc = incl - run;   // exclusive prefix of this lane's chunk
  for (int j = 0; j < C; ++j) {
    int k = lane * C + j;
    if (k < nw) {
      acc += (double)((w(k) + 1e-5f) / sum);
      cdf[k + 1] = (float)acc;
    }
  }
  if (lane == 0) cdf[0] = 0.f;
  for (int k = lane; k < Nb; k += 64) bins[k] = bin(k);
}

// searchsorted(cdf, u, right=True): number of entries <= u.
__device__ __forceinline__ int upper_bound(const float* cdf, int Nb, float u) {
  int lo = 0, hi = Nb;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int Nb, float u, int* ind_out) {
  int ind = upper_bound(cdf, Nb, u);
  *ind_out = ind;
  int below = ind - 1 < 0 ? 0 : ind - 1;
  int above = ind > Nb - 1 ? Nb - 1 : ind;
  float c0 = cdf[below], c1 = cdf[above];
  float b0 = bins[below], b1 = bins[above];
  float denom = c1 - c0;
  if (denom < 1e-5f) denom = 1.f;
  float t = (u - c0) / denom;
  return b0 + t * (b1 - b0);
}

__global__ __launch_bounds__(WAVES * 64) void sample_pdf_k(const float* __restrict__ bins_g,
                                                           const float* __restrict__ weights_g,
                                                           const float* __restrict__ u_g, int64_t u_stride,
                                                           int64_t B, int Nb, int Nf, float* __restrict__ samples,
                                                           int64_t* __restrict__ inds) {
  __shared__ PdfLds lds;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + wv;
  if (b >= B) return;
  float* cdf = lds.cdf[wv];
  float* bins = lds.bins[wv];
  const float* wrow = weights_g + b * (Nb - 1);
  const float* brow = bins_g + b * Nb;
  build_cdf(cdf, bins, Nb, lane, [&](int k) { return wrow[k]; }, [&](int k) { return brow[k]; });
  __builtin_amdgcn_wave_barrier();
  const float* urow = u_g + b * u_stride;
  for (int k = lane; k < Nf; k += 64) {
    int ind;
    float s = invert_cdf(cdf, bins, Nb, urow[k], &ind);
    samples[b * Nf + k] = s;
    if (inds) inds[b * Nf + k] = ind;
  }
}

struct ResampleLds {
  float cdf[WAVES][MAX_NB];
  float bins[WAVES][MAX_NB];
  float all[WAVES][MAX_ALL];
};

// z_mid -> sample_pdf(z_mid, weights[1:-1]) -> sort(cat[z, z_samples]) and std(z_samples), one ray/wave.
__global__ __launch_bounds__(WAVES * 64) void resample_k(const float* __restrict__ z_g,
                                                         const float* __restrict__ weights_g,
                                                         const float* __restrict__ u_g, int64_t u_stride, int64_t B,
                                                         int Nc, int Nf, float* __restrict__ z_fine,
                                                         float* __restrict__ z_std, float* __restrict__ samples,
                                                         int64_t* __restrict__ inds, int use_rng, CnRngK rngk) {
  __shared__ ResampleLds lds;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + wv;
  if (b >= B) return;
  float* cdf = lds.cdf[wv];
  float* bins = lds.bins[wv];
  float* all = lds.all[wv];
  const float* zrow = z_g + b * Nc;
  const float* wrow = weights_g + b * Nc;
  const int Nb = Nc - 1;
  build_cdf(cdf, bins, Nb, lane, [&](int k) { return wrow[k + 1]; },
            [&](int k) { return .5f * (zrow[k + 1] + zrow[k]); });
  for (int k = lane; k < Nc; k += 64) all[k] = zrow[k];
  __builtin_amdgcn_wave_barrier();
  const float* urow = use_rng ? nullptr : u_g + b * u_stride;
  const CnRngDev rng(rngk);
  double s1 = 0.0;
  for (int k = lane; k < Nf; k += 64) {
    int ind;
    // (use_rng: u is generated here — element (row0 + b, k) of the global [rays, Nf] stream, rng.hpp)
    float s = invert_cdf(cdf, bins, Nb, use_rng ? rng.uniform(b, Nf, k) : urow[k], &ind);
    all[Nc + k] = s;
    s1 += (double)s;
    if (samples) samples[b * Nf + k] = s;
    if (inds) inds[b * Nf + k] = ind;
  }
  __builtin_amdgcn_wave_barrier();
  // population std of the new samples (R:415), two-pass in fp64
  const double mean = wave_sum(s1) / (double)Nf;
  double s2 = 0.0;
  for (int k = lane; k < Nf; k += 64) {
    double d = (double)all[Nc + k] - mean;
    s2 += d * d;
  }
  s2 = wave_sum(s2);
  if (lane == 0) z_std[b] = (float)sqrt(s2 / (double)Nf);
  // ascending sort of the Nc+Nf depths (values only, R:399)
  const int n = Nc + Nf;
  float* out = z_fine + b * n;
  if (n <= 256) {
    // bitonic network over 256 slots (4 per lane, slot e = 64 r + lane, padding = +inf) entirely in registers: 36 compare-exchange
    // steps, 33 of them one cross-lane exchange per register (partner lane ^ j), 3 between registers of the same lane.  ~700
    // instructions per ray instead of the ~2700 of an all-pairs rank sort (which made this kernel VALU-bound: 4 waves per SIMD x
    // 5.5 us at 4096 rays).  Values only, so equal keys need no tie rule.  INPUTS MUST BE NaN-FREE: fminf / fmaxf return the
    // non-NaN operand, so a NaN depth would be dropped and its neighbour duplicated (torch.sort keeps NaNs, last).  The depths
    // are convex combinations of finite near / far bounds and bin edges, so a NaN here means NaN rays, which R:417-419 reports.
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (64 * r + lane) < n ? all[64 * r + lane] : __builtin_inff();
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j >= 1; j >>= 1) {
        if (j >= 64) {
          const int dr = j >> 6;            // partner register r ^ dr, same lane
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((r & dr) == 0) {
              const bool asc = (((64 * r) & k) == 0);
              const float lo = fminf(v[r], v[r | dr]), hi = fmaxf(v[r], v[r | dr]);
              v[r] = asc ? lo : hi;
              v[r | dr] = asc ? hi : lo;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float o = __shfl_xor(v[r], j, 64);
            const bool asc = (((64 * r + lane) & k) == 0);
            const bool keep_min = ((lane & j) == 0) == asc;
            v[r] = keep_min ? fminf(v[r], o) : fmaxf(v[r], o);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (64 * r + lane < n) out[64 * r + lane] = v[r];
    return;
  }
  // (more than 256 depths per ray: all-pairs rank sort; ties keep input order)
  for (int e = lane; e < n; e += 64) {
    const float x = all[e];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float y = all[j];
      rank += (y < x) || (y == x && j < e);
    }
    out[rank] = x;
  }
}

}  // namespace

extern "C" int cnerf_coarse_z(const float* rays, int ray_stride, int64_t B, int Nc, const float* t_vals,
                              const float* t_rand, int lindisp, float* z, void* stream) {
  if (!rays || !t_vals || !z || B < 0 || Nc <= 0 || ray_stride < 8) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  int64_t n = B * Nc;
  hipLaunchKernelGGL(coarse_z_k, dim3((unsigned)cn_div_up(n, 256)), dim3(256), 0, cn_stream(stream), rays, ray_stride,
                     B, Nc, t_vals, t_rand, lindisp, z, 0.f, 0.f, 0, CnRngK{});
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// coarse depths for the rays of a camera (near / far are its constants); used by cnerf_render_fwd_cam
int cn_coarse_z_cam(float near, float far, int64_t B, int Nc, const float* t_vals, const float* t_rand, int lindisp,
                    float* z, hipStream_t st) {
  if (!t_vals || !z || B < 0 || Nc <= 0) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  int64_t n = B * Nc;
  hipLaunchKernelGGL(coarse_z_k, dim3((unsigned)cn_div_up(n, 256)), dim3(256), 0, st, (const float*)nullptr, 0, B, Nc,
                     t_vals, t_rand, lindisp, z, near, far, 0, CnRngK{});
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_row_stride,
                                int64_t B, int Nb, int Nf, float* samples, int64_t* inds, void* stream) {
  if (!bins || !weights || !u || !samples || B < 0 || Nb < 2 || Nf <= 0) return CNERF_E_ARG;
  if (Nb > MAX_NB) return CNERF_E_UNSUPPORTED;
  if (B == 0) return CNERF_OK;
  hipLaunchKernelGGL(sample_pdf_k, dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, cn_stream(stream), bins,
                     weights, u, u_row_stride, B, Nb, Nf, samples, inds);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_resample(const float* z, const float* weights, const float* u, int64_t u_row_stride, int64_t B,
                              int Nc, int Nf, float* z_fine, float* z_std, float* samples, int64_t* inds,
                              void* stream) {
  if (!z || !weights || !u || !z_fine || !z_std || B < 0 || Nc < 3 || Nf <= 0) return CNERF_E_ARG;
  if (Nc - 1 > MAX_NB || Nc + Nf > MAX_ALL) return CNERF_E_UNSUPPORTED;
  if (B == 0) return CNERF_OK;
  hipLaunchKernelGGL(resample_k, dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, cn_stream(stream), z,
                     weights, u, u_row_stride, B, Nc, Nf, z_fine, z_std, samples, inds, 0, CnRngK{});
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

// ---- the same two entry points with the uniform streams generated in-kernel (rng.hpp) -----------------------------------------
namespace {
__global__ void uniform_rng_k(CnRngK rngk, int64_t rows, int cols, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int64_t b = idx / cols;
  out[idx] = CnRngDev(rngk).uniform(b, cols, (int)(idx - b * cols));
}
}  // namespace

extern "C" int cnerf_uniform_rng(const cnerf_rng* rng, int64_t rows, int cols, float* out, void* stream) {
  if (!rng || !out || rows < 0 || cols <= 0) return CNERF_E_ARG;
  if (rows == 0) return CNERF_OK;
  hipLaunchKernelGGL(uniform_rng_k, dim3((unsigned)cn_div_up(rows * cols, 256)), dim3(256), 0, cn_stream(stream), cn_rng_arg(rng),
                     rows, cols, out);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_coarse_z_rng(const float* rays, int ray_stride, int64_t B, int Nc, const float* t_vals, const cnerf_rng* rng,
                                  int lindisp, float* z, void* stream) {
  if (!rays || !t_vals || !z || !rng || B < 0 || Nc <= 0 || ray_stride < 8) return CNERF_E_ARG;
  if (B == 0) return CNERF_OK;
  hipLaunchKernelGGL(coarse_z_k, dim3((unsigned)cn_div_up(B * Nc, 256)), dim3(256), 0, cn_stream(stream), rays, ray_stride, B, Nc,
                     t_vals, (const float*)nullptr, lindisp, z, 0.f, 0.f, 1, cn_rng_arg(rng));
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}

extern "C" int cnerf_resample_rng(const float* z, const float* weights, const cnerf_rng* rng, int64_t B, int Nc, int Nf,
                                  float* z_fine, float* z_std, float* samples, int64_t* inds, void* stream) {
  if (!z || !weights || !rng || !z_fine || !z_std || B < 0 || Nc < 3 || Nf <= 0) return CNERF_E_ARG;
  if (Nc - 1 > MAX_NB || Nc + Nf > MAX_ALL) return CNERF_E_UNSUPPORTED;
  if (B == 0) return CNERF_OK;
  hipLaunchKernelGGL(resample_k, dim3((unsigned)cn_div_up(B, WAVES)), dim3(WAVES * 64), 0, cn_stream(stream), z, weights,
                     (const float*)nullptr, (int64_t)0, B, Nc, Nf, z_fine, z_std, samples, inds, 1, cn_rng_arg(rng));
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
