// sin and cos of the positional encoding's arguments (Embedder, H:24-45: 2^l x for l = 0..9 — |a| up to a few thousand rad at the
// scene scales of the reference's datasets) with one range reduction for the pair.
//   k = rint(a 2/pi);  r = a - k pi/2 by ONE fp64 FMA (pi/2 to 53 bits: the reduced argument is exact to 1e-10 for |a| < 2^20),
//   rounded once to fp32;  sin r and cos r by degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4] (fp32 FMAs; fitting error
//   5e-10 / 2e-10);  quadrant fix-up from k mod 4.
// ~25 instructions.  The library's sincosf is generic (any argument, Payne-Hanek for the large ones, ~140 instructions with its
// branches): in the MLP forward, whose VALU time is not hidden behind the fp32 MFMAs (mlp_common.hpp), the 21 calls per lane of a
// 32-point tile were 1.4 % of the kernel.  It remains the path for |a| >= 2^20, infinities and NaN.
// Accuracy over the encodings' argument ranges (scripts/sincos_check.py: a numpy restatement with emulated FMAs, 2 M samples per
// frequency): max |error| 7.0e-8 against a float64 evaluation (the reference's own CPU torch.sin / torch.cos: 3.6e-8), never more
// than one ulp (6.0e-8) from them, bit-equal on 83 % of the samples.  Built with -DCN_SINCOS_OWN; without it the library call.
#pragma once
#include <math.h>

__device__ __forceinline__ void cn_sincos(float a, float* sn, float* cs) {
#ifndef CN_SINCOS_OWN   // (until the A/B on the MI355X is in: scripts/gpu_r05_sincos.sh)
  sincosf(a, sn, cs);
#else
  if (!(fabsf(a) < 1048576.f)) {   // huge, inf, nan
    sincosf(a, sn, cs);
    return;
  }
  const double ad = (double)a;
  const double kd = __builtin_rint(ad * 0.63661977236758134308);           // 2 / pi
  const float r = (float)__builtin_fma(kd, -1.57079632679489661923, ad);   // |r| <= pi/4 (+ rounding)
  const int q = (int)kd;
  const float z = r * r;
  float ps = __builtin_fmaf(z, 2.6021755274996394e-06f, -0.00019828711810987443f);
  ps = __builtin_fmaf(z, ps, 0.008333305828273296f);
  ps = __builtin_fmaf(z, ps, -0.1666666716337204f);
  const float s = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.4463804948027246e-05f, -0.0013887588866055012f);
  pc = __builtin_fmaf(z, pc, 0.04166664928197861f);
  pc = __builtin_fmaf(z, pc, -0.5f);
  const float c = __builtin_fmaf(z, pc, 1.f);
  const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
#endif
}
