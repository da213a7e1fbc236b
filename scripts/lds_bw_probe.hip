// LDS read bandwidth per CU with the ring's access pattern (4 waves, each ds_read_b128 = 64 lanes x 16 contiguous bytes), and the
// same while a bf16 MFMA stream runs in the same waves: what the A-operand reads of the bf16x3 ring kernels can get.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int READS_PER_MFMA_X2, bool MFMA>
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  f32x16 q[2];
  for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) q[t][r] = lane * 1e-3f;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  const unsigned char* base = lds + lane * 16;
  u32x4 ring[8];
  for (int r = 0; r < 8; ++r) ring[r] = u32x4{0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 48; ++j) {
      if (MFMA) q[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), q[j & 1], 0, 0, 0);
      // READS_PER_MFMA_X2 / 2 reads per MFMA; a value is consumed 8 reads after it was issued (the kernel: two tiles = 12 MFMAs)
      if (READS_PER_MFMA_X2 >= 2 || (j & 1) == 0) {
#pragma unroll
        for (int r = 0; r < (READS_PER_MFMA_X2 + 1) / 2; ++r) {
          const int n = (READS_PER_MFMA_X2 >= 2 ? j * (READS_PER_MFMA_X2 / 2) + r : j / 2);
          acc ^= ring[n & 7];
          ring[n & 7] = *reinterpret_cast<const u32x4*>(base + (n % 96) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int r = 0; r < 8; ++r) acc ^= ring[r];
  float s = 0; for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += q[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ (unsigned)s;
}
template <int R2, bool MFMA>
void run(unsigned* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4, iters = 2000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<R2, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL((probe<R2, MFMA>), dim3(blocks), dim3(256), 96 * 1024, 0, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<R2, MFMA>), dim3(blocks), dim3(256), 96 * 1024, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int reads_per_iter = (R2 >= 2) ? 48 * ((R2 + 1) / 2) : 24;
  const double bytes = (double)blocks * 4 * iters * reads_per_iter * 1024.0;      // per wave 1 KiB per read
  const double clk = ms * 1e-3 * 2.4e9 * 256 / ((blocks + 255) / 256) / 1.0;      // CU-cycles at 2.4 GHz per round of blocks
  printf("reads per MFMA %.1f, MFMA %d: %.3f ms  LDS read %.1f B/clk/CU (at 2.4 GHz), %.1f cycles per 48-MFMA K-step per wave\n", R2 / 2.0, (int)MFMA, ms,
         bytes / (ms * 1e-3 * 2.4e9 * 256), ms * 1e-3 * 2.4e9 / (4.0 * iters));
}
int main() {
  unsigned* out; (void)hipMalloc(&out, 4 * 256 * 1024);
  run<1, false>(out); run<2, false>(out); run<4, false>(out); run<8, false>(out);
  run<1, true>(out); run<2, true>(out); run<4, true>(out);
  return 0;
}
