// Latency from issuing ONE 16-byte-per-lane buffer store (or load) to s_waitcnt vmcnt(0) returning, with the machine
// otherwise busy (all CUs run the same loop) — coalesced (1 KiB contiguous per wave) vs row-scattered (32 rows x 32 B,
// the former point-major stash pattern).   hipcc --offload-arch=gfx950 -O3 -o scripts/storeack_probe scripts/storeack_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int MODE>   // 0 coalesced store, 1 scattered store, 2 coalesced load (L2 miss: streaming), 3 load L2-hot
__global__ __launch_bounds__(64) void probe(unsigned long long* cyc, float* big, int iters, int gap) {
  const int lane = threadIdx.x;
  rsrc_t os = __builtin_amdgcn_make_buffer_rsrc((void*)(big + (size_t)blockIdx.x * (16u << 20)), 0, 0x7fffffff, 0x00027000);
  const int voff = MODE == 1 ? ((lane & 31) * 10240 + (lane >> 5) * 16) : lane * 16;
  unsigned long long tot = 0;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int i = 0; i < iters; ++i) {
    const int so = MODE == 3 ? 0 : MODE == 1 ? (i & 63) * 32 + (i >> 6) * 327680 : i * 1024;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE <= 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), os, voff, so, 0);
    else v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os, voff, so, 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    tot += t1 - t0;
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
  }
  if (lane == 0) cyc[blockIdx.x] = tot / iters;
  if (v[0] == 123.456f) big[0] = v[1];
}

template <int MODE>
void run(const char* name, unsigned long long* cyc, float* big, int blocks, int gap) {
  std::vector<unsigned long long> h(blocks);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(64), 0, 0, cyc, big, 2000, gap);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-28s waves=%4d gap=%d: %.0f cycles issue->vmcnt(0)\n", name, blocks, gap, s / blocks);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned long long* cyc; float* big;
  (void)hipMalloc(&cyc, 4096 * 8); (void)hipMalloc(&big, (size_t)1024 * (64u << 20));
  for (int blocks : {1, 1024}) for (int gap : {0, 4}) {
    run<0>("store coalesced 1 KiB", cyc, big, blocks, gap);
    run<1>("store scattered 32 x 32 B", cyc, big, blocks, gap);
    run<2>("load streaming", cyc, big, blocks, gap);
    run<3>("load L2-hot", cyc, big, blocks, gap);
  }
  return 0;
}
