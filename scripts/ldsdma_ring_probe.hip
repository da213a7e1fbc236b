// Probe of the LDS-DMA ring protocol of csrc/mlp_fwd_bf.hip (shared-panel kernel): 4 waves, a 4-slot ring of PIECES x 1 KiB,
// K-step s+2 DMA'd (each wave a quarter of the pieces) while K-step s is read, one vmcnt + barrier per K-step.  Every wave
// checks every 16-byte chunk it reads against the source pattern.  Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_ring_probe
// scripts/ldsdma_ring_probe.hip ; run: ./ldsdma_ring_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma1k(const i32x4& rs, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rs),
                 "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

template <int PIECES, int WORK>
__global__ __launch_bounds__(256) void probe_k(const unsigned* src, int nsteps, unsigned bytes, unsigned* bad, unsigned* sink) {
  constexpr int SLOT = PIECES * 1024, PW = PIECES / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long ba = (unsigned long long)src;
  const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((ba >> 32) & 0xffff)),
                    __builtin_amdgcn_readfirstlane((int)bytes), 0x00027000};
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds);
  const int rd16 = (m * 2 + hh) * 16;
  auto dma = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int i = w + 4 * j;
      dma1k(rs, lds0 + (unsigned)(slot * SLOT + i * 1024), lane * 16, (s * PIECES + i) * 1024);
    }
  };
  unsigned nbad = 0, acc = 0;
  dma(0, 0);
  dma(1, 1);
  __builtin_amdgcn_s_waitcnt(0x0f70 | (PW & 15) | ((PW >> 4) << 14));
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    if (s + 2 < nsteps) dma(s + 2, (s + 2) & 3);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + (s & 3) * SLOT + i * 1024 + rd16);
      const unsigned want = ((unsigned)(s * PIECES + i) * 1024 + rd16) / 4;     // source dword index = its value
#pragma unroll
      for (int k = 0; k < 4; ++k) nbad += v[k] != want + k;
      for (int r = 0; r < WORK; ++r) acc = acc * 1664525u + v[r & 3];            // some time between K-steps
    }
    if (s + 2 < nsteps) __builtin_amdgcn_s_waitcnt(0x0f70 | (PW & 15) | ((PW >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }
  atomicAdd(bad, nbad);
  if (acc == 12345u) sink[0] = acc;
}

template <int PIECES, int WORK>
void run(const unsigned* src, int nsteps, unsigned bytes, unsigned* bad, unsigned* sink, int blocks) {
  const size_t ldsb = 4 * PIECES * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe_k<PIECES, WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((probe_k<PIECES, WORK>), dim3(blocks), dim3(256), ldsb, 0, src, nsteps, bytes, bad, sink);
  unsigned h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("PIECES %2d WORK %3d blocks %5d: mismatching dwords %u  (%s)\n", PIECES, WORK, blocks, h, hipGetErrorString(hipGetLastError()));
}

int main() {
  const int nsteps = 160;
  const unsigned bytes = nsteps * 24 * 1024;
  unsigned* h = (unsigned*)malloc(bytes);
  for (unsigned i = 0; i < bytes / 4; ++i) h[i] = i;
  unsigned *src, *bad, *sink;
  hipMalloc(&src, bytes); hipMalloc(&bad, 4); hipMalloc(&sink, 4);
  hipMemcpy(src, h, bytes, hipMemcpyHostToDevice);
  for (int blocks : {1, 256, 6144}) {
    run<4, 0>(src, nsteps, bytes, bad, sink, blocks);
    run<4, 64>(src, nsteps, bytes, bad, sink, blocks);
    run<8, 0>(src, nsteps, bytes, bad, sink, blocks);
    run<8, 64>(src, nsteps, bytes, bad, sink, blocks);
    run<16, 0>(src, nsteps, bytes, bad, sink, blocks);
    run<24, 16>(src, nsteps, bytes, bad, sink, blocks);
  }
  return 0;
}
