// Which fp32 MFMA shape does the power cap favour?  Sustained rate of v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 (same
// nominal 256 flop / cycle / CU-SIMD... 64 flop/cycle/SIMD), 1 wave64 per SIMD, 8 independent accumulators, with CONSTANT and with
// RANDOM operands (the data the MLP kernels see): the 32x32x2 shape moves 16 accumulator registers per 4096 flop, 16x16x4 four per
// 2048.   hipcc --offload-arch=gfx950 -O3 scripts/mfma_f32_shapes_probe.hip -o scripts/mfma_f32_shapes_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
}

template <bool RANDOM>
__global__ __launch_bounds__(64) void probe32(float* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  f32x16 acc[8];
  float x[16], y[16];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = RANDOM ? rnd(s) : 1.0f;
  for (int k = 0; k < 16; ++k) { x[k] = RANDOM ? rnd(s) : 1e-3f; y[k] = RANDOM ? rnd(s) * 0.06f : 1e-3f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k], y[(k + t) & 15], acc[t], 0, 0, 0);
  }
  float r = 0;
  for (int t = 0; t < 8; ++t) for (int q = 0; q < 16; ++q) r += acc[t][q];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <bool RANDOM>
__global__ __launch_bounds__(64) void probe16(float* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  f32x4 acc[8];
  float x[16], y[16];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = RANDOM ? rnd(s) : 1.0f;
  for (int k = 0; k < 16; ++k) { x[k] = RANDOM ? rnd(s) : 1e-3f; y[k] = RANDOM ? rnd(s) * 0.06f : 1e-3f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[k], y[(k + t) & 15], acc[t], 0, 0, 0);
  }
  float r = 0;
  for (int t = 0; t < 8; ++t) for (int q = 0; q < 4; ++q) r += acc[t][q];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <class K>
void run(const char* name, K kern, double flop_per_mfma, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 1024;
  for (int rep = 0; rep < 2; ++rep) {
    const int iters = rep == 0 ? 64 : (flop_per_mfma > 3000 ? 3000 : 6000);     // ~25 ms at peak
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-34s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles-at-2.4GHz per MFMA)\n", name, ms,
                    (double)blocks * iters * 128 * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 128));
  }
}

int main() {
  float* out; hipMalloc(&out, 4 * 64 * 1024);
  for (int pass = 0; pass < 2; ++pass) {
    run("32x32x2 f32, constant operands", probe32<false>, 4096.0, out);
    run("32x32x2 f32, random operands", probe32<true>, 4096.0, out);
    run("16x16x4 f32, constant operands", probe16<false>, 2048.0, out);
    run("16x16x4 f32, random operands", probe16<true>, 2048.0, out);
  }
  return 0;
}
