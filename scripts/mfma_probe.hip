// Sustained v_mfma_f32_32x32x2_f32 rate at this kernel family's occupancy (1 wave64 per SIMD, 8 independent
// accumulators, ~10 ms of work): the practical MFMA ceiling (DVFS included) the fused MLP kernels are compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void probe(float* out, int iters, float a, float b) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + t;
  float x = a + threadIdx.x * 1e-6f, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 64 * 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1024, 4096, 24576}) {
    int iters = 4096 * 1024 / blocks * 2;
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, 64, 1e-3f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, iters, 1e-3f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * iters * 32 * 2.0 * 32 * 32 * 2;
    printf("blocks=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", blocks, iters, ms, flop / ms / 1e9);
  }
  return 0;
}
