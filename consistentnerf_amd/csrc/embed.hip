// Stand-alone positional encoding (Embedder.embed, H:15-45) for callers that use the embedder object
// directly; the render path never materialises encodings (they are generated inside mlp_fwd.hip).
#include "common.hpp"

namespace {
__global__ void embed_k(const float* __restrict__ x, int64_t M, int L, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int ch = 3 + 6 * L;
  const float v[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  float* o = out + i * ch;
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  float f = 1.f;
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float sn, cs;
      sincosf(v[d] * f, &sn, &cs);
      o[3 + 6 * l + d] = sn;
      o[3 + 6 * l + 3 + d] = cs;
    }
    f *= 2.f;
  }
}
}  // namespace

extern "C" int cnerf_embed(const float* x, int64_t M, int L, float* out, void* stream) {
  if (!x || !out || M < 0 || L < 0 || L > 16) return CNERF_E_ARG;
  if (M == 0) return CNERF_OK;
  hipLaunchKernelGGL(embed_k, dim3((unsigned)cn_div_up(M, 256)), dim3(256), 0, cn_stream(stream), x, M, L, out);
  CN_CHECK_LAUNCH();
  return CNERF_OK;
}
