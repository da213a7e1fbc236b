// Cost of one 16-byte-per-lane load, by addressing mode, for a wave issuing v_mfma_f32_32x32x2_f32 back to back
// (one wave per SIMD).  N loads are placed behind the first N of every 16 MFMAs; results are consumed at the end.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/opcost2_probe scripts/opcost2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { NONE = 0, G_V64 = 1, G_SADDR = 2, B_OFFEN = 3, B_TID = 4, B_LDS = 5, DS128 = 6, B_OFF = 7, B_STORE = 8, B_STORE_LD = 9, DS_WRITE = 10, B_LDS_M0 = 11, B_LDS4 = 12, B_LDS4_M0 = 13, B_LDS4_STREAM = 14, B_OFFEN_STREAM = 15 };

template <int KIND, int N>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, const float* wts, int iters, float a,
                                             float b, float* big) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
  f32x4 sink[N > 0 ? N : 1];
  for (int n = 0; n < (N > 0 ? N : 1); ++n) sink[n] = f32x4{0, 0, 0, 0};
  const float* base;                                // wave-uniform (made so explicitly: SGPR operands below)
  {
    const unsigned long long ba0 = (unsigned long long)(wts + w * 65536);
    const unsigned lo0 = __builtin_amdgcn_readfirstlane((int)(ba0 & 0xffffffffu));
    const unsigned hi0 = __builtin_amdgcn_readfirstlane((int)(ba0 >> 32));
    base = (const float*)(((unsigned long long)hi0 << 32) | lo0);
  }
  const float* pa = base + lane * 4;                // per-lane 64-bit address
  const unsigned voff = lane * 16;                  // per-lane 32-bit byte offset
  i32x4 rsrc, rsrc_tid;
  {
    const unsigned long long ba = (unsigned long long)base;
    const int lo = __builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu));
    const int hi = __builtin_amdgcn_readfirstlane((int)(ba >> 32));
    rsrc = i32x4{lo, hi & 0xffff, (int)0x7fffffff, 0x00027000};
    rsrc_tid = i32x4{lo, (hi & 0xffff) | (16 << 16), (int)0x7fffffff, 0x00027000 | (1 << 23)};
  }
  i32x4 orsrc;   // this wave's private 32 MiB slice of the output buffer (streaming writes, like the stash)
  {
    const unsigned long long ba = (unsigned long long)(big + ((size_t)blockIdx.x * 4 + w) * (8u << 20));
    const int lo = __builtin_amdgcn_readfirstlane((int)(ba & 0xffffffffu));
    const int hi = __builtin_amdgcn_readfirstlane((int)(ba >> 32));
    orsrc = i32x4{lo, hi & 0xffff, (int)0x7fffffff, 0x00027000};
  }
  const unsigned ldsaddr = (unsigned)(size_t)(lds) + w * 16384 + lane * 16;   // LDS byte address of this lane's slot
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    const unsigned so = ((i * 16) & 255) * 1024;    // wave-uniform byte offset of this iteration's first load
    const float* sbase = base + so / 4;
    if (KIND == B_LDS || KIND == B_LDS4 || KIND == B_LDS4_STREAM) asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane((int)(w * 16384 + 32768))));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        const int s = 4 * j + t;
        if (s < N) {
          if (KIND == G_V64) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(sink[s]) : "v"(pa + so / 4), "n"(s * 128));
          if (KIND == G_SADDR) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(sink[s]) : "v"(voff), "s"(sbase), "n"(s * 128));
          if (KIND == B_OFFEN) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(sink[s]) : "v"(voff), "s"(rsrc), "s"(so), "n"(s * 128));
          if (KIND == B_TID) asm volatile("buffer_load_dwordx4 %0, off, %1, %2 offset:%3" : "=v"(sink[s]) : "s"(rsrc_tid), "s"(so), "n"(s * 128));
          if (KIND == B_OFF) asm volatile("buffer_load_dwordx4 %0, off, %1, %2 offset:%3" : "=v"(sink[s]) : "s"(rsrc), "s"(so), "n"(s * 128));
          if (KIND == B_LDS) asm volatile("buffer_load_dword %0, %1, %2 offen offset:%3 lds" ::"v"(voff / 4), "s"(rsrc), "s"(so), "n"(s * 128) : "memory");
          if (KIND == B_LDS_M0) asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen offset:%3 lds" ::"v"(voff / 4), "s"(rsrc), "s"(so), "n"(s * 128), "s"(__builtin_amdgcn_readfirstlane((int)(w * 16384 + 32768 + s * 256))) : "memory");
          if (KIND == B_LDS4) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(so), "n"((s & 3) * 1024) : "memory");
          if (KIND == B_LDS4_STREAM) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(orsrc), "s"((unsigned)i * 16384u + (unsigned)(s >> 2) * 4096u), "n"((s & 3) * 1024) : "memory");
          if (KIND == B_OFFEN_STREAM) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(sink[s]) : "v"(voff), "s"(orsrc), "s"((unsigned)i * 16384u + (unsigned)(s >> 2) * 4096u), "n"((s & 3) * 1024));
          if (KIND == B_LDS4_M0) asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(so), "n"((s & 3) * 1024), "s"(__builtin_amdgcn_readfirstlane((int)(w * 16384 + 32768 + (s & 3) * 1024))) : "memory");
          if (KIND == B_STORE || (KIND == B_STORE_LD && (s & 1))) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4" ::"v"(sink[s]), "v"(voff), "s"(orsrc), "s"(so + (unsigned)(i >> 4) * 262144u), "n"(s * 128) : "memory");
          if (KIND == B_STORE_LD && !(s & 1)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(sink[s]) : "v"(voff), "s"(rsrc), "s"(so), "n"(s * 128));
          if (KIND == DS_WRITE) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ldsaddr), "v"(sink[s]), "n"(s * 1024) : "memory");
          if (KIND == DS128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[s]) : "v"(ldsaddr), "n"(s * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int n = 0; n < (N > 0 ? N : 1); ++n) s += sink[n][0] + sink[n][1] + sink[n][2] + sink[n][3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

static float* g_big;
template <int KIND, int N>
void run(const char* name, float* out, unsigned long long* cyc, const float* wts) {
  const int blocks = 256, iters = 2000;
  (void)hipFuncSetAttribute((const void*)probe<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  std::vector<unsigned long long> h(blocks * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<KIND, N>), dim3(blocks), dim3(256), 128 * 1024, 0, out, cyc, wts, iters, 1e-3f, 1e-3f, g_big);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s N=%d: launch failed\n", name, N); return; }
  }
  (void)hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  const double per = s / h.size() / (iters * 16.0);
  printf("%-14s N=%2d per 16 MFMA: %.2f cycles/MFMA  (+%.1f cycles per load)\n", name, N, per, N ? (per - 64.04) * 16 / N : 0.0);
}

int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : -1;
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *out, *wts; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
  (void)hipMalloc(&wts, 4 * 65536 * 4 + 65536); (void)hipMemset(wts, 0, 4 * 65536 * 4 + 65536);
  (void)hipMalloc(&g_big, (size_t)1024 * (32u << 20));   // 32 GiB: 1024 waves x 32 MiB
  if (which < 0 || which == B_STORE) { run<B_STORE, 1>("buffer store", out, cyc, wts); run<B_STORE, 2>("buffer store", out, cyc, wts); run<B_STORE, 4>("buffer store", out, cyc, wts); run<B_STORE, 8>("buffer store", out, cyc, wts); }
  if (which < 0 || which == B_STORE_LD) { run<B_STORE_LD, 4>("store+load", out, cyc, wts); run<B_STORE_LD, 8>("store+load", out, cyc, wts); run<B_STORE_LD, 16>("store+load", out, cyc, wts); }
  if (which < 0 || which == DS_WRITE) { run<DS_WRITE, 4>("ds_write_b128", out, cyc, wts); run<DS_WRITE, 8>("ds_write_b128", out, cyc, wts); }
  if (which < 0 || which == NONE) run<NONE, 0>("none", out, cyc, wts);
  if (which < 0 || which == G_V64) { run<G_V64, 4>("global v64", out, cyc, wts); run<G_V64, 8>("global v64", out, cyc, wts); run<G_V64, 16>("global v64", out, cyc, wts); }
  if (which < 0 || which == G_SADDR) { run<G_SADDR, 4>("global saddr", out, cyc, wts); run<G_SADDR, 8>("global saddr", out, cyc, wts); run<G_SADDR, 16>("global saddr", out, cyc, wts); }
  if (which < 0 || which == B_OFFEN) { run<B_OFFEN, 4>("buffer offen", out, cyc, wts); run<B_OFFEN, 8>("buffer offen", out, cyc, wts); run<B_OFFEN, 16>("buffer offen", out, cyc, wts); }
  if (which < 0 || which == B_OFF) run<B_OFF, 8>("buffer off", out, cyc, wts);
  if (which < 0 || which == B_TID) { run<B_TID, 4>("buffer tid", out, cyc, wts); run<B_TID, 8>("buffer tid", out, cyc, wts); run<B_TID, 16>("buffer tid", out, cyc, wts); }
  if (which < 0 || which == B_LDS) { run<B_LDS, 8>("buffer->lds", out, cyc, wts); run<B_LDS, 16>("buffer->lds", out, cyc, wts); }
  if (which < 0 || which == B_LDS_M0) { run<B_LDS_M0, 8>("buf->lds m0 each", out, cyc, wts); run<B_LDS_M0, 16>("buf->lds m0 each", out, cyc, wts); }
  if (which < 0 || which == B_LDS4) { run<B_LDS4, 4>("buf->lds x4", out, cyc, wts); run<B_LDS4, 8>("buf->lds x4", out, cyc, wts); run<B_LDS4, 16>("buf->lds x4", out, cyc, wts); }
  if (which < 0 || which == B_LDS4_M0) { run<B_LDS4_M0, 4>("buf->lds x4 m0 each", out, cyc, wts); run<B_LDS4_M0, 8>("buf->lds x4 m0 each", out, cyc, wts); run<B_LDS4_M0, 16>("buf->lds x4 m0 each", out, cyc, wts); }
  if (which < 0 || which == B_LDS4_STREAM) { run<B_LDS4_STREAM, 1>("lds-dma x4 stream", out, cyc, wts); run<B_LDS4_STREAM, 2>("lds-dma x4 stream", out, cyc, wts); run<B_LDS4_STREAM, 4>("lds-dma x4 stream", out, cyc, wts); run<B_LDS4_STREAM, 8>("lds-dma x4 stream", out, cyc, wts); }
  if (which < 0 || which == B_OFFEN_STREAM) { run<B_OFFEN_STREAM, 1>("vgpr load stream", out, cyc, wts); run<B_OFFEN_STREAM, 2>("vgpr load stream", out, cyc, wts); run<B_OFFEN_STREAM, 4>("vgpr load stream", out, cyc, wts); run<B_OFFEN_STREAM, 8>("vgpr load stream", out, cyc, wts); }
  if (which < 0 || which == DS128) { run<DS128, 4>("ds_read_b128", out, cyc, wts); run<DS128, 8>("ds_read_b128", out, cyc, wts); run<DS128, 16>("ds_read_b128", out, cyc, wts); }
  return 0;
}
