// Does a buffer STORE stream delay the consumption of buffer LOADS that follow it (vmcnt retires in order)?
// One wave per SIMD; per 16 MFMAs the wave issues 4 A-operand loads that the MFMAs two groups later consume
// (so the compiler waits with vmcnt(N)) and NS 16-byte-per-lane streaming stores.  Reports cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/opcost3_probe scripts/opcost3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int NS, int AHEAD>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, const float* wts, int iters, float* big) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 0x7fffffff, 0x00027000);
  rsrc_t os = __builtin_amdgcn_make_buffer_rsrc((void*)(big + ((size_t)blockIdx.x * 4) * (8u << 20)), 0, 0x7fffffff, 0x00027000);
  const int voff = lane * 16, wo = w * (32 << 20);
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
  f32x4 a[AHEAD + 1][4];
  for (int d = 0; d < AHEAD; ++d)
    for (int t = 0; t < 4; ++t) a[d][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (d * 4 + t) * 1024, 0));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i += AHEAD + 1) {
#pragma unroll
    for (int d = 0; d <= AHEAD; ++d) {
      const int so = (((i + d + AHEAD) * 4) & 1023) * 1024;
      f32x4 (&cur)[4] = a[d];
      f32x4 (&nxt)[4] = a[(d + AHEAD) % (AHEAD + 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][j], 1e-3f, acc[t], 0, 0, 0);
          const int s = 4 * j + t;
          if (j == 3) nxt[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + t * 1024, 0));
          if (j < 3 && s < NS)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{acc[0][s], acc[1][s], acc[2][s], acc[3][s]}), os, voff,
                                                   wo + ((i + d) & 8191) * 4096 + s * 1024, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

static float* g_big;
template <int NS, int AHEAD>
void run(float* out, unsigned long long* cyc, const float* wts) {
  const int blocks = 256, iters = 2400;
  std::vector<unsigned long long> h(blocks * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<NS, AHEAD>), dim3(blocks), dim3(256), 0, 0, out, cyc, wts, iters, g_big);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
  }
  (void)hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("stores/16 MFMA = %d, A prefetch distance = %d groups: %.2f cycles/MFMA\n", NS, AHEAD, s / h.size() / (iters * 16.0));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *out, *wts; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
  (void)hipMalloc(&wts, 8 << 20); (void)hipMemset(wts, 0, 8 << 20);
  (void)hipMalloc(&g_big, (size_t)256 * (128u << 20));
  run<0, 1>(out, cyc, wts); run<1, 1>(out, cyc, wts); run<2, 1>(out, cyc, wts); run<4, 1>(out, cyc, wts);
  run<0, 2>(out, cyc, wts); run<1, 2>(out, cyc, wts); run<2, 2>(out, cyc, wts); run<4, 2>(out, cyc, wts);
  run<1, 3>(out, cyc, wts); run<2, 3>(out, cyc, wts); run<4, 3>(out, cyc, wts);
  return 0;
}
