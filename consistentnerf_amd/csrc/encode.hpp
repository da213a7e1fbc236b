// Positional encoding of one point into an LDS tile (shared by the fp32 and the bf16-plane forward kernels).
#pragma once
#include "mlp_common.hpp"

// gamma(v) of this lane's point into the LDS tile T[m][0..chp): half-wave hh takes the frequencies l = hh, hh+2, ...
// (one sincos per coordinate: both channels of the pair), half-wave 0 the identity channels, half-wave 1 the zero
// padding.  Channel order H:24-45: [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(..)].  With `pre`, the
// channels are copied from an already-embedded row instead (NeRF.forward(x), H:107-109).
__device__ __forceinline__ void encode(float* T, const float (&v)[3], int L, int ch, int chp, int m, int hh,
                                       const float* __restrict__ pre) {
  auto put = [&](int k, float x) { T[enc_off(m, k >> 2) + (k & 3)] = x; };
  if (pre != nullptr) {
    for (int k = hh; k < chp; k += 2) put(k, k < ch ? pre[k] : 0.f);
    return;
  }
  if (hh == 0) {
    put(0, v[0]); put(1, v[1]); put(2, v[2]);
  } else {
    for (int k = ch; k < chp; ++k) put(k, 0.f);
  }
  float f = hh ? 2.f : 1.f;
  for (int l = hh; l < L; l += 2) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float sn, cs;
      sincosf(v[d] * f, &sn, &cs);   // one range reduction for the pair (same values as sinf / cosf)
      put(3 + 6 * l + d, sn);
      put(3 + 6 * l + 3 + d, cs);
    }
    f *= 4.f;
  }
}

