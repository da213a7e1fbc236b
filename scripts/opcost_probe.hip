// What does a memory instruction cost a wave that is otherwise issuing v_mfma_f32_32x32x2_f32 back to back?
// One wave per SIMD (256-thread workgroups, one per CU); every 16 MFMAs the wave issues N loads of one kind whose
// results are only consumed at the very end.  Reports cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/opcost_probe scripts/opcost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { NONE = 0, G128 = 1, G64 = 2, G32 = 3, L128 = 4, L64 = 5, L32 = 6, G128_USE = 7, VALU4 = 8 };

template <int KIND, int N>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, const float* wts, int iters, float a,
                                             float b) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
  f32x4 sink[N > 0 ? N : 1];
  for (int n = 0; n < (N > 0 ? N : 1); ++n) sink[n] = f32x4{0, 0, 0, 0};
  float x = a, y = b;
  const float* pa = wts + lane * 4 + (threadIdx.x >> 6) * 65536;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (KIND == G128_USE) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sink[(4 * j + t) % (N > 0 ? N : 1)][j] + x, y, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t], 0, 0, 0);
        const int s = 4 * j + t;
        if (s < N) {
          const float* p = pa + ((i * 16 + s) & 255) * 256;
          float* l = lds + ((lane * 4 + (i * 16 + s) * 256) & 8191);
          if (KIND == G128 || KIND == G128_USE) sink[s] = *reinterpret_cast<const f32x4*>(p);
          if (KIND == G64) { const f32x2 v = *reinterpret_cast<const f32x2*>(p); sink[s][0] = v[0]; sink[s][1] = v[1]; }
          if (KIND == G32) sink[s][0] = *p;
          if (KIND == L128) sink[s] = *reinterpret_cast<volatile f32x4*>(l);
          if (KIND == L64) { const f32x2 v = *reinterpret_cast<volatile f32x2*>(l); sink[s][0] = v[0]; sink[s][1] = v[1]; }
          if (KIND == L32) sink[s][0] = *reinterpret_cast<volatile float*>(l);
          if (KIND == VALU4) { sink[s][0] = __builtin_fmaf(sink[s][0], a, b); sink[s][1] = __builtin_fmaf(sink[s][1], a, b);
                               sink[s][2] = __builtin_fmaf(sink[s][2], a, b); sink[s][3] = __builtin_fmaf(sink[s][3], a, b); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int n = 0; n < (N > 0 ? N : 1); ++n) s += sink[n][0] + sink[n][1] + sink[n][2] + sink[n][3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, float* out, unsigned long long* cyc, const float* wts) {
  const int blocks = 256, iters = 2000;
  hipFuncSetAttribute((const void*)probe<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  std::vector<unsigned long long> h(blocks * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<KIND, N>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, wts, iters, 1e-3f, 1e-3f);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  const double per = s / h.size() / (iters * 16.0);
  printf("%-10s N=%d per 16 MFMA: %.2f cycles/MFMA  (+%.1f cycles per instruction)\n", name, N, per,
         N ? (per - 66.3) * 16 / N : 0.0);
}

int main() {
  float *out, *wts; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  hipMalloc(&wts, 4 * 65536 * 4 + 4096); hipMemset(wts, 0, 4 * 65536 * 4 + 4096);
  run<NONE, 0>("none", out, cyc, wts);
  run<G128, 4>("global128", out, cyc, wts); run<G128, 8>("global128", out, cyc, wts); run<G128, 16>("global128", out, cyc, wts);
  run<G64, 4>("global64", out, cyc, wts); run<G64, 8>("global64", out, cyc, wts);
  run<G32, 4>("global32", out, cyc, wts); run<G32, 16>("global32", out, cyc, wts);
  run<L128, 4>("lds128", out, cyc, wts); run<L128, 8>("lds128", out, cyc, wts); run<L128, 16>("lds128", out, cyc, wts);
  run<L64, 8>("lds64", out, cyc, wts); run<L32, 16>("lds32", out, cyc, wts);
  run<G128_USE, 4>("g128+use", out, cyc, wts); run<G128_USE, 8>("g128+use", out, cyc, wts);
  run<VALU4, 4>("4xv_fma", out, cyc, wts); run<VALU4, 8>("4xv_fma", out, cyc, wts);
  return 0;
}
