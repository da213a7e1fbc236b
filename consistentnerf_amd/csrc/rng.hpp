// Counter-based uniform streams for the two torch.rand draws of render_rays — the stratified jitter t_rand (R:376) and the
// resampling positions u (H:227) — generated INSIDE coarse_z_k / resample_k instead of by a generator launch whose output
// round-trips HBM.  Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's CUDA streams as well), one block per
// element: key = seed ^ a fixed constant (a stream disjoint from every torch.rand call on the same seed), counter =
// (element index [64 bits], stream offset [64 bits]) with element = (row0 + ray) * cols + col — the index of the value in the
// GLOBAL [total_rays, cols] draw, so a shard of a batch (row0 = its first global row) sees exactly the rows the unsharded call
// sees, for free.  value = (x0 >> 8) * 2^-24: the 24-bit grid on [0, 1) of ATen's CPU torch.rand for float32.
// oracle/philox.py restates this in numpy (pinned on the Random123 known-answer vectors); tests compare bit for bit.
#pragma once
#include "common.hpp"

struct CnRngK {
  uint64_t seed, offset;
  const uint64_t* dev;   // non-null: {seed, base offset} live in device memory (hipGraph replays); `offset` is added to the base
  int64_t row0;
};

static inline CnRngK cn_rng_arg(const cnerf_rng* r) {
  CnRngK k;
  k.seed = r->seed; k.offset = r->offset; k.dev = r->state_dev; k.row0 = r->row0;
  return k;
}

__device__ __forceinline__ uint32_t cn_philox_x0(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

struct CnRngDev {
  uint64_t key, off;
  int64_t row0;
  __device__ __forceinline__ explicit CnRngDev(const CnRngK& r) {
    const uint64_t seed = r.dev ? r.dev[0] : r.seed;
    key = seed ^ 0x636e6572665f726eull;               // "cnerf_rn"
    off = r.offset + (r.dev ? r.dev[1] : 0ull);
    row0 = r.row0;
  }
  __device__ __forceinline__ float uniform(int64_t row, int cols, int col) const {
    const uint64_t e = (uint64_t)(row0 + row) * (uint64_t)cols + (uint64_t)col;
    return (float)(cn_philox_x0(e, off, key) >> 8) * 5.9604644775390625e-8f;
  }
};
