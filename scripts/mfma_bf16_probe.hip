// Sustained v_mfma_f32_32x32x16_bf16 rate at the bf16x3 kernels' occupancy (1 wave64 per SIMD): the practical ceiling (DVFS
// included) of the opt-in arithmetic.  Variants: NACC independent accumulators round-robin (16 = no dependency at all, 2 =
// two alternating chains, 1 = one dependent chain), and the same with ~3 VALU instructions between MFMAs (the plane splits).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int VALU, bool RANDOM>
__global__ __launch_bounds__(64) void probe(float* out, int iters, unsigned a, unsigned b) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + t;
  u32x4 x = {a + threadIdx.x, a, a, a}, y = {b, b, b + threadIdx.x, b};
  u32x4 xs[4], ys[4];      // RANDOM: four pseudo-random operand sets per lane, cycled (realistic toggle activity -> realistic power)
  {
    unsigned h = 0x9e3779b9u * (threadIdx.x + 64 * blockIdx.x + 1) + a;
    for (int k = 0; k < 4; ++k)
      for (int e = 0; e < 4; ++e) {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        xs[k][e] = (h & 0x807f807fu) | 0x3f003f00u;          // two bf16 in [0.5, 1) with random sign and mantissa
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        ys[k][e] = (h & 0x807f807fu) | 0x3f003f00u;
      }
  }
  float v0 = threadIdx.x * 1e-3f, v1 = 1.0001f, v2 = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16 / NACC * 4; ++j)
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        if (RANDOM) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xs[(j + t) & 3]), __builtin_bit_cast(bf16x8, ys[(j * 3 + t) & 3]), acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[t], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < VALU; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = v0;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC, int VALU, bool RANDOM>
void run(float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096 * 1024 / blocks * 4;
  hipLaunchKernelGGL((probe<NACC, VALU, RANDOM>), dim3(blocks), dim3(64), 0, 0, out, 16, 0x3f803f80u, 0x3f803f80u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, VALU, RANDOM>), dim3(blocks), dim3(64), 0, 0, out, iters, 0x3f803f80u, 0x3f803f80u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)blocks * iters * 64.0;                    // MFMAs per wave: iters * (16/NACC*4) * NACC = 64 iters
  const double flop = nm * 2.0 * 32 * 32 * 16;
  printf("%s blocks=%5d NACC=%2d VALU/MFMA=%d  %.3f ms  %.1f TFLOP/s bf16 = %.1f TFLOP/s fp32-equivalent at 6 products (%.1f cycles per MFMA at 2.4 GHz)\n",
         RANDOM ? "random operands" : "constant operands", blocks, NACC, VALU, ms, flop / ms / 1e9, flop / ms / 1e9 / 6, ms * 1e-3 * 2.4e9 / (nm / blocks * ((blocks + 1023) / 1024)) );
}
int main() {
  float* out; hipMalloc(&out, 4 * 64 * 32768);
  for (int blocks : {4096}) {
    run<16, 0, false>(out, blocks); run<1, 0, false>(out, blocks); run<16, 3, false>(out, blocks); run<2, 6, false>(out, blocks);
    run<16, 0, true>(out, blocks); run<2, 0, true>(out, blocks); run<1, 0, true>(out, blocks);
    run<16, 3, true>(out, blocks); run<2, 3, true>(out, blocks); run<2, 6, true>(out, blocks);
    run<16, 0, true>(out, blocks);      // (again: the clock has settled)
  }
  return 0;
}
