// -DCN_TIMING (experiment builds only, scripts/ktiming.py): per-wave cycle accounting of a kernel's phases with
// s_memtime.  Slots: 0 encode/prologue, 1 barrier waits, 2 GEMM, 3 park, 4 heads, 5 total, 6/7 realtime (100 MHz)
// begin/end, 8 HW_ID, 9 XCC_ID, 10.. [start, end) stamps of the first 15 GEMM phases.
#pragma once
#ifdef CN_TIMING
#define CN_TSLOTS 40
static __device__ unsigned long long cn_tbuf[CN_TSLOTS * 65536];   // one per translation unit (no device linking)
#define CN_TIMING_ACCESSOR(NAME)                                                                            \
  extern "C" int NAME(unsigned long long* host, int64_t n) {                                                \
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_tbuf), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1; \
  }
#define CN_TINIT(WPB)                                                                   \
  unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0};                                       \
  unsigned long long t_ev[30];                                                          \
  int t_n = 0;                                                                          \
  const unsigned t_slot = (blockIdx.y * gridDim.x + blockIdx.x) * (WPB) + (threadIdx.x >> 6);                      \
  const unsigned long long t_rt0 = wall_clock64();                                      \
  unsigned long long t_last = __builtin_readcyclecounter();                             \
  const unsigned long long t_begin = t_last;
#define CN_T(i)                                                                         \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    const unsigned long long t_now = __builtin_readcyclecounter();                      \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    t_acc[i] += t_now - t_last;                                                         \
    if (i == 2 && t_n < 30) { t_ev[t_n++] = t_last; t_ev[t_n++] = t_now; }              \
    t_last = t_now;                                                                     \
  }
#define CN_TEND                                                                         \
  {                                                                                     \
    if ((threadIdx.x & 63) == 0 && t_slot < 65536) {                                    \
      unsigned long long* o = cn_tbuf + (size_t)t_slot * CN_TSLOTS;                     \
      t_acc[5] = t_last - t_begin;                                                      \
      for (int i = 0; i < 6; ++i) o[i] = t_acc[i];                                      \
      o[6] = t_rt0;                                                                     \
      o[7] = wall_clock64();                                                            \
      o[8] = __builtin_amdgcn_s_getreg(63492);  /* HW_ID */                             \
      o[9] = __builtin_amdgcn_s_getreg(63508);  /* XCC_ID */                            \
      for (int i = 0; i < 30; ++i) o[10 + i] = i < t_n ? t_ev[i] : 0;                   \
    }                                                                                   \
  }
#else
#define CN_TINIT(WPB)
#define CN_T(i)
#define CN_TEND
#endif
