// Do two waves on one SIMD overlap MFMA with VALU / LDS / VMEM work?  (gfx950 issue-port probe)
// One 512-thread workgroup per CU: waves w and w+4 land on the same SIMD (checked via HW_ID).  Waves 0-3 run role A,
// waves 4-7 role B; each wave times a fixed amount of its own work with s_memtime.  Comparing "alone" with
// "together" gives the slowdown each kind of work inflicts on the other.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/coissue_probe scripts/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { IDLE = 0, MFMA = 1, VALU = 2, LDSR = 3, VMEM = 4, MFMA_LD = 5, TRANS = 6, MFMA_N32 = 7, MFMA_N48 = 8, VALU_PRIO = 9, MFMA_N56 = 10 };

__device__ __forceinline__ float run_role(int role, int iters, const float* __restrict__ wts, float* lds, int lane,
                                          float a, float b) {
  float s = 0.f;
  if (role == MFMA) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else if (role == MFMA_N32 || role == MFMA_N48 || role == MFMA_N56) {   // MFMA followed by s_nop: leaves issue slots
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
    if (role == MFMA_N32) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15"); asm volatile("s_nop 15");
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    } else if (role == MFMA_N48) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15");
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    } else {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 7");
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else if (role == MFMA_LD) {   // the fused-MLP inner loop: 4 x 16-byte A loads + 1 LDS read per 16 MFMAs, 1 group ahead
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = lane * 1e-3f + t;
    const float* pa = wts + lane * 4;
    f32x4 r0[4], r1[4];
    for (int t = 0; t < 4; ++t) r0[t] = *reinterpret_cast<const f32x4*>(pa + t * 256);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + lane * 4);
    for (int i = 0; i < iters; i += 2) {
      const float* p1 = pa + ((i + 1) & 255) * 2048;
      for (int t = 0; t < 4; ++t) r1[t] = *reinterpret_cast<const f32x4*>(p1 + t * 256);
      f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256 + 256) & 8191));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0[t][j], b0[j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const float* p2 = pa + ((i + 2) & 255) * 2048;
      for (int t = 0; t < 4; ++t) r0[t] = *reinterpret_cast<const f32x4*>(p2 + t * 256);
      b0 = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256 + 512) & 8191));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1[t][j], b1[j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else if (role == VALU || role == VALU_PRIO) {
    if (role == VALU_PRIO) __builtin_amdgcn_s_setprio(3);
    float x[8];
    for (int t = 0; t < 8; ++t) x[t] = a + t + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 32; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = __builtin_fmaf(x[t], a, b);
    }
    for (int t = 0; t < 8; ++t) s += x[t];
  } else if (role == TRANS) {
    float x = a + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __sinf(x) + b;
    }
    s = x;
  } else if (role == LDSR) {
    f32x4 v = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x4 q = *reinterpret_cast<volatile f32x4*>(lds + ((lane * 4 + (i * 16 + j) * 256) & 8191));
        v += q;
      }
    }
    s = v[0] + v[1] + v[2] + v[3];
  } else if (role == VMEM) {
    f32x4 v = {0, 0, 0, 0};
    const float* pa = wts + lane * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v += *reinterpret_cast<const f32x4*>(pa + (((i * 16 + j) * 7) & 2047) * 256);
    }
    s = v[0] + v[1] + v[2] + v[3];
  }
  return s;
}

__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, unsigned* hw, const float* wts,
                                             int roleA, int itA, int roleB, int itB, float a, float b) {
  extern __shared__ float lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 1e-4f;
  __syncthreads();
  const int role = w < 4 ? roleA : roleB, it = w < 4 ? itA : itB;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const float s = run_role(role, it, wts, lds, lane, a, b);
  __builtin_amdgcn_sched_barrier(0);
  out[(blockIdx.x * 8 + w) * 64 + lane] = s;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) {
    cyc[blockIdx.x * 8 + w] = t1 - t0;
    hw[blockIdx.x * 8 + w] = __builtin_amdgcn_s_getreg(63492);
  }
}

static const char* NAME[] = {"idle", "mfma", "valu", "ldsr", "vmem", "mfma+ld", "trans", "mfma_n32", "mfma_n48", "valu_prio", "mfma_n56"};

int main() {
  const int blocks = 256;
  float *out, *wts; unsigned long long* cyc; unsigned* hw;
  hipMalloc(&out, blocks * 8 * 64 * 4); hipMalloc(&cyc, blocks * 8 * 8); hipMalloc(&hw, blocks * 8 * 4);
  hipMalloc(&wts, 2048 * 256 * 4 + 4096); hipMemset(wts, 0, 2048 * 256 * 4 + 4096);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  std::vector<unsigned long long> h(blocks * 8); std::vector<unsigned> hh(blocks * 8);
  // iteration counts sized for ~1M cycles of solo work each
  auto iters = [](int role) { return (role == MFMA || role >= MFMA_N32 && role != VALU_PRIO) ? 1000 : role == VALU_PRIO ? 1000 : role == MFMA_LD ? 1000 : role == VALU ? 1000 : role == TRANS ? 4000
                                     : role == LDSR ? 4000 : role == VMEM ? 1500 : 0; };
  const int combos[][2] = {{MFMA, IDLE}, {MFMA_LD, IDLE}, {VALU, IDLE}, {TRANS, IDLE}, {LDSR, IDLE}, {VMEM, IDLE},
                           {MFMA, MFMA}, {MFMA_LD, MFMA_LD}, {MFMA, VALU}, {MFMA_LD, VALU}, {MFMA, TRANS}, {MFMA, LDSR},
                           {MFMA, VMEM}, {MFMA_LD, VMEM}, {VALU, VALU}, {MFMA, MFMA_LD},
                           {MFMA_N32, IDLE}, {MFMA_N48, IDLE}, {MFMA_N56, IDLE}, {MFMA_N32, VALU}, {MFMA_N48, VALU}, {MFMA_N56, VALU},
                           {MFMA_N48, MFMA_N48}, {MFMA, VALU_PRIO}, {VALU_PRIO, MFMA}, {MFMA_N48, TRANS}, {MFMA_N48, LDSR}, {MFMA_N48, VMEM}};
  for (auto& c : combos) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 100 * 1024, 0, out, cyc, hw, wts, c[0], iters(c[0]), c[1],
                         iters(c[1]), 1e-3f, 1e-3f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hh.data(), hw, blocks * 8 * 4, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0; int same = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 4; ++w) {
        sa += h[b * 8 + w]; sb += h[b * 8 + 4 + w];
        same += ((hh[b * 8 + w] >> 4) & 3) == ((hh[b * 8 + 4 + w] >> 4) & 3);
      }
    printf("A=%-8s B=%-8s  cycles A %9.0f  B %9.0f   (waves w,w+4 on the same SIMD: %d/%d)\n", NAME[c[0]], NAME[c[1]],
           sa / (blocks * 4), sb / (blocks * 4), same, blocks * 4);
  }
  return 0;
}
